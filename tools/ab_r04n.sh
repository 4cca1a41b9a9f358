#!/bin/bash
# GPU box: the small shards of the data-parallel jobs (SMAC at the 64 threads one of 8 GPUs owns, the north star at 512) on the
# final library against the library of the commit before the second half of round 4 (libmappo_hip_OLD.so), alternating on
# one box -- do the leaner kernels cost anything where the step is launch-latency-bound?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_n
mkdir -p $OUT
cd $REPO
OLD=$REPO/on-policy_amd/lib/libmappo_hip_OLD.so
for i in 1 2 3; do
  MAPPO_HIP_LIB=$OLD timeout 300 python bench.py --workload smac --threads 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/smac64_old.jsonl
  timeout 300 python bench.py --workload smac --threads 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/smac64_new.jsonl
done
for i in 1 2; do
  MAPPO_HIP_LIB=$OLD timeout 300 python bench.py --workload ns --threads 512 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/ns512_old.jsonl
  timeout 300 python bench.py --workload ns --threads 512 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/ns512_new.jsonl
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_n/"
for name in ("smac64_old", "smac64_new", "ns512_old", "ns512_new"):
    print(name, [json.loads(l)["ms_per_step"] for l in open(out + name + ".jsonl") if l.startswith("{")])
PY
