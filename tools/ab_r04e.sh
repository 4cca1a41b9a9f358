#!/bin/bash
# GPU box: tanh with 7 instead of 16 instructions (default) against rounds 2-3's form (libmappo_hip_POLY.so): parity tests, then
# alternating microbenchmarks and north-star steps on one box.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_e
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py tests/test_gpu_gru_seq.py tests/test_gpu_runners.py -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
POLY=$REPO/on-policy_amd/lib/libmappo_hip_POLY.so
for i in 1 2 3; do
  timeout 200 python tools/bench_mlp.py --sequential --reps 7 >> $OUT/mlp_new.jsonl 2>&1
  MAPPO_HIP_LIB=$POLY timeout 200 python tools/bench_mlp.py --sequential --reps 7 >> $OUT/mlp_poly.jsonl 2>&1
done
for i in 1 2; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_ns_new.jsonl
  MAPPO_HIP_LIB=$POLY timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_ns_poly.jsonl
done
for w in cfg3 ns_rnn smac; do
  timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_other_new.jsonl
  MAPPO_HIP_LIB=$POLY timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_other_poly.jsonl
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_e/"
for name in ("mlp_new", "mlp_poly"):
    rows = [json.loads(l) for l in open(out + name + ".jsonl") if l.startswith("{")]
    for din in (384, 48):
        print(name, "din", din, "fwd_ms", [r["fwd_ms"] for r in rows if r["din"] == din], "bwd_ms", [r["bwd_ms"] for r in rows if r["din"] == din])
for name in ("bench_ns_new", "bench_ns_poly", "bench_other_new", "bench_other_poly"):
    for l in open(out + name + ".jsonl"):
        if l.startswith("{"):
            d = json.loads(l)
            print(name, d["config"]["workload"][:40], d["ms_per_step"], "fwd", d["roofline"]["launch_ms"], d["roofline"]["frac"])
PY
