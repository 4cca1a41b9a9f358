#!/bin/bash
# GPU box: (a) K7 with a row's probabilities in registers for heads of <= 8 actions + packed head dot products in the chain,
# against the library of the previous commit (libmappo_hip_OLD.so); (b) the version-3 forward for the narrow actor inputs
# too (MAPPO_MLP_FLAGS=16) now that its loop is leaner.  Alternating on one box.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_i
mkdir -p $OUT
cd $REPO
OLD=$REPO/on-policy_amd/lib/libmappo_hip_OLD.so
timeout 600 python -m pytest tests/test_gpu_fused_loss.py tests/test_gpu_action_spaces.py tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py tests/test_gpu_parity.py -q > $OUT/tests_new.log 2>&1; tail -2 $OUT/tests_new.log
for w in ns cfg3 ns; do
  MAPPO_HIP_LIB=$OLD timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_old.jsonl
  timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_new.jsonl
  MAPPO_MLP_FLAGS=16 timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_f16.jsonl
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_i/"
for name in ("bench_old", "bench_new", "bench_f16"):
    for l in open(out + name + ".jsonl"):
        if l.startswith("{"):
            d = json.loads(l); print(name, d["config"]["workload"][:36], d["ms_per_step"], "fwd", d["roofline"].get("launch_ms"), "bwd", d["roofline_mlp_backward"]["launch_ms"])
PY
