#!/bin/bash
# GPU box: the version-3 forward with a shortened last chunk (template parameter NQL: only the groups of 8 columns that hold
# real data) for the narrow inputs (MAPPO_MLP_FLAGS=16) against the loader / compute kernel (default below 129 columns), same
# library, alternating on one box; device parity of the forward first.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_m
mkdir -p $OUT
cd $REPO
MAPPO_MLP_FLAGS=16 timeout 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py -q > $OUT/tests_f16.log 2>&1; tail -2 $OUT/tests_f16.log
timeout 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py -q > $OUT/tests_f0.log 2>&1; tail -2 $OUT/tests_f0.log
for i in 1 2 3; do
  timeout 200 python tools/bench_mlp.py --sequential --reps 7 >> $OUT/mlp_f0.jsonl 2>&1
  MAPPO_MLP_FLAGS=16 timeout 200 python tools/bench_mlp.py --sequential --reps 7 >> $OUT/mlp_f16.jsonl 2>&1
done
for w in ns cfg3 cfg2 ns; do
  timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_f0.jsonl
  MAPPO_MLP_FLAGS=16 timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_f16.jsonl
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_m/"
for name in ("mlp_f0", "mlp_f16"):
    rows = [json.loads(l) for l in open(out + name + ".jsonl") if l.startswith("{")]
    for din in (384, 48):
        print(name, "din", din, "fwd_ms", [r["fwd_ms"] for r in rows if r["din"] == din])
for name in ("bench_f0", "bench_f16"):
    for l in open(out + name + ".jsonl"):
        if l.startswith("{"):
            d = json.loads(l); print(name, d["config"]["workload"][:40], d["ms_per_step"], "fwd", d["roofline"].get("launch_ms"))
PY
