#!/bin/bash
# round 3: backward chain v2 -- correctness on the device, then A/B against v1 (MAPPO_MLP_FLAGS=4) inside one call
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"
OUT=gpurun_out/r03_c2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py -q -s > $OUT/tests.log 2>&1
tail -25 $OUT/tests.log
echo "--- v2"; timeout 300 python tools/bench_mlp.py --reps 7 2>&1 | tail -2 | tee $OUT/mlp_v2.jsonl
echo "--- v1"; MAPPO_MLP_FLAGS=4 timeout 300 python tools/bench_mlp.py --reps 7 2>&1 | tail -2 | tee $OUT/mlp_v1.jsonl
echo "--- prof v2"; bash tools/profile_mlp.sh --reps 5 2>&1 | tail -14 | tee $OUT/mlp_prof_v2.txt
timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee $OUT/bench_ns.json
