#!/bin/bash
# GPU box: the direct-to-LDS first-layer weight-gradient kernel with two slots per wave and two workgroups per CU (flags 32)
# against the shipped form (four slots, one workgroup per CU): parity, then alternating runs on one box + counters.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_f
mkdir -p $OUT
cd $REPO
MAPPO_MLP_FLAGS=32 timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py -q > $OUT/tests_32.log 2>&1; tail -2 $OUT/tests_32.log
for i in 1 2 3; do
  MAPPO_MLP_FLAGS=0 timeout 200 python tools/bench_mlp.py --sequential --reps 7 --din 384 >> $OUT/mlp_0.jsonl 2>&1
  MAPPO_MLP_FLAGS=32 timeout 200 python tools/bench_mlp.py --sequential --reps 7 --din 384 >> $OUT/mlp_32.jsonl 2>&1
done
for f in 0 32; do
  MAPPO_MLP_FLAGS=$f timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_ns_$f.jsonl
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_f/"
for f in (0, 32):
    rows = [json.loads(l) for l in open(out + "mlp_%d.jsonl" % f) if l.startswith("{")]
    print("flags", f, "bwd_ms", [r["bwd_ms"] for r in rows], "fwd_ms", [r["fwd_ms"] for r in rows])
    for l in open(out + "bench_ns_%d.jsonl" % f):
        if l.startswith("{"):
            d = json.loads(l); print("   ns", d["ms_per_step"], "bwd", d["roofline_mlp_backward"]["launch_ms"], d["roofline_mlp_backward"]["frac"])
PY
