#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"
OUT=gpurun_out/r03_c6; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py -q > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
echo "--- new"; timeout 300 python tools/bench_mlp.py --reps 7 2>&1 | tail -2 | tee $OUT/mlp_new.jsonl
echo "--- old fwd"; MAPPO_MLP_FLAGS=16 timeout 300 python tools/bench_mlp.py --reps 7 2>&1 | tail -2 | tee $OUT/mlp_oldfwd.jsonl
echo "--- sequential new"; timeout 300 python tools/bench_mlp.py --reps 7 --sequential 2>&1 | tail -2 | tee $OUT/mlp_new_seq.jsonl
echo "--- prof"; bash tools/profile_mlp.sh --reps 5 2>&1 | tail -12 | tee $OUT/mlp_prof.txt
timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee $OUT/bench_ns.json
