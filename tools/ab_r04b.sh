#!/bin/bash
# GPU box, second A / B of round 4: (1) the rollout graph (tests + config-3 end to end), (2) the K9 forward in three forms on
# one box: version 3 with 8 waves (flags 16: every aligned width), with 12 waves (24), loader / compute kernel (4).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_b
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_rollout_graph.py tests/test_gpu_mpe_end_to_end.py tests/test_gpu_mlp.py -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 600 python tools/cfg3_end_to_end.py --out $OUT/cfg3_end_to_end.json > $OUT/cfg3_end_to_end.log 2>&1; tail -1 $OUT/cfg3_end_to_end.log | cut -c1-700
for i in 1 2 3; do
  for f in 16 24 4; do
    MAPPO_MLP_FLAGS=$f timeout 200 python tools/bench_mlp.py --sequential --reps 7 --din 384 48 436 152 >> $OUT/mlp_$f.jsonl 2>&1
  done
done
MAPPO_MLP_FLAGS=24 timeout 200 python tools/bench_mlp.py --sequential --reps 3 --din 384 --stamps > $OUT/stamps_12w.log 2>&1
for f in 0 8 4; do
  MAPPO_MLP_FLAGS=$f timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_ns_$f.jsonl
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_b/"
for f in (16, 24, 4):
    rows = [json.loads(l) for l in open(out + "mlp_%d.jsonl" % f) if l.startswith("{")]
    for din in (384, 436, 152, 48):
        print("flags", f, "din", din, "fwd_ms", [r["fwd_ms"] for r in rows if r["din"] == din])
for f in (0, 8, 4):
    for l in open(out + "bench_ns_%d.jsonl" % f):
        if l.startswith("{"):
            d = json.loads(l)
            print("ns flags", f, d["ms_per_step"], "fwd", d["roofline"]["launch_ms"], d["roofline"]["frac"])
PY
