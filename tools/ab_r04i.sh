#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_i
mkdir -p $OUT
cd $REPO
for i in 1 2 3; do
  for m in 1 0; do
    MAPPO_CONCURRENT_NETS=$m timeout 300 python bench.py --workload smac --threads 64 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/smac64_$m.jsonl
  done
done
for m in 1 0; do
  MAPPO_CONCURRENT_NETS=$m timeout 400 python bench.py --workload ns_rnn --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/nsrnn_$m.jsonl
  MAPPO_CONCURRENT_NETS=$m timeout 300 python bench.py --workload smac --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/smac512_$m.jsonl
  MAPPO_CONCURRENT_NETS=$m timeout 300 python bench.py --workload ns_rnn --threads 512 --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/nsrnn512_$m.jsonl
done
python - <<'PY'
import json, os, glob
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_i/"
for f in sorted(glob.glob(out + "*.jsonl")):
    print(os.path.basename(f), [json.loads(l)["ms_per_step"] for l in open(f) if l.startswith("{")])
PY
