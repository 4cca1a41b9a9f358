#!/bin/bash
# GPU box: A / B of the K9 forward kernels inside ONE box (they differ by +-10 % between boxes): version 3 (default) against
# the loader / compute kernel (MAPPO_MLP_FLAGS=4).  Device parity first, then alternating microbenchmarks at the north-star
# shapes (rows in memory order, as the device sampler hands them over), cycle stamps, then the north-star step itself.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_fwd3
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py tests/test_gpu_sampler_indices.py -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for i in 1 2 3; do
  MAPPO_MLP_FLAGS=0 timeout 200 python tools/bench_mlp.py --sequential --reps 7 >> $OUT/mlp_v3.jsonl 2>&1
  MAPPO_MLP_FLAGS=4 timeout 200 python tools/bench_mlp.py --sequential --reps 7 >> $OUT/mlp_loaders.jsonl 2>&1
done
MAPPO_MLP_FLAGS=0 timeout 200 python tools/bench_mlp.py --sequential --reps 3 --stamps > $OUT/stamps_v3.log 2>&1
for i in 1 2; do
  MAPPO_MLP_FLAGS=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_ns_v3.jsonl
  MAPPO_MLP_FLAGS=4 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_ns_loaders.jsonl
done
for w in cfg2 cfg3 smac ns_rnn; do
  MAPPO_MLP_FLAGS=0 timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_other_v3.jsonl
  MAPPO_MLP_FLAGS=4 timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_other_loaders.jsonl
done
python - <<'PY'
import json, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_fwd3/"
for name in ("mlp_v3", "mlp_loaders"):
    rows = [json.loads(l) for l in open(out + name + ".jsonl") if l.startswith("{")]
    for din in sorted({r["din"] for r in rows}):
        print(name, "din", din, "fwd_ms", [r["fwd_ms"] for r in rows if r["din"] == din], "bwd_ms", [r["bwd_ms"] for r in rows if r["din"] == din])
for name in ("bench_ns_v3", "bench_ns_loaders", "bench_other_v3", "bench_other_loaders"):
    for l in open(out + name + ".jsonl"):
        if l.startswith("{"):
            d = json.loads(l)
            print(name, d["config"]["workload"][:40], d["ms_per_step"], "fwd", d["roofline"]["launch_ms"], d["roofline"]["frac"])
PY
