#!/bin/bash
# GPU box: the first-layer weight-gradient kernels with the upper 8 rows of their LDS tiles shifted by 32 floats (no two-way
# bank conflict between the half-waves of an operand read) against the library of the previous commit, alternating on one box.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_k
mkdir -p $OUT
cd $REPO
OLD=$REPO/on-policy_amd/lib/libmappo_hip_OLD.so
timeout 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py -q > $OUT/tests_new.log 2>&1; tail -2 $OUT/tests_new.log
for i in 1 2 3; do
  MAPPO_HIP_LIB=$OLD timeout 200 python tools/bench_mlp.py --sequential --reps 7 >> $OUT/mlp_old.jsonl 2>&1
  timeout 200 python tools/bench_mlp.py --sequential --reps 7 >> $OUT/mlp_new.jsonl 2>&1
done
for w in ns smac ns; do
  MAPPO_HIP_LIB=$OLD timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_old.jsonl
  timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_new.jsonl
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_k/"
for name in ("mlp_old", "mlp_new"):
    rows = [json.loads(l) for l in open(out + name + ".jsonl") if l.startswith("{")]
    for din in (384, 48):
        print(name, "din", din, "fwd_ms", [r["fwd_ms"] for r in rows if r["din"] == din], "bwd_ms", [r["bwd_ms"] for r in rows if r["din"] == din])
for name in ("bench_old", "bench_new"):
    for l in open(out + name + ".jsonl"):
        if l.startswith("{"):
            d = json.loads(l); print(name, d["config"]["workload"][:36], d["ms_per_step"], "fwd", d["roofline"].get("launch_ms"), "bwd", d["roofline_mlp_backward"]["launch_ms"])
PY
