#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"
OUT=gpurun_out/r03_c3; mkdir -p $OUT
echo "--- stamps v2 (2 waves/SIMD)"; timeout 300 python tools/bench_mlp.py --reps 3 --stamps --din 48 2>&1 | tail -12 | tee $OUT/stamps_v2.txt
echo "--- stamps v2 (1 wave/SIMD)"; MAPPO_MLP_FLAGS=8 timeout 300 python tools/bench_mlp.py --reps 3 --stamps --din 48 2>&1 | tail -12 | tee $OUT/stamps_v2_1w.txt
echo "--- 1 wave/SIMD timing"; MAPPO_MLP_FLAGS=8 timeout 300 python tools/bench_mlp.py --reps 7 --din 48 2>&1 | tail -1 | tee $OUT/mlp_v2_1w.jsonl
