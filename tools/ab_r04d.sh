#!/bin/bash
# GPU box: rollout refinements (critic branch of the graph on a side stream, one-launch uniform draws, K14 sampling kernel):
# tests + config-3 end to end.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_d
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_fused_loss.py tests/test_gpu_rollout_graph.py tests/test_gpu_mpe_end_to_end.py tests/test_gpu_runners.py tests/test_gpu_scripts.py tests/test_gpu_action_spaces.py -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 600 python tools/cfg3_end_to_end.py --out $OUT/cfg3_end_to_end.json > $OUT/cfg3_end_to_end.log 2>&1; tail -1 $OUT/cfg3_end_to_end.log | cut -c1-900
timeout 600 python tools/cfg3_end_to_end.py --algorithm_name rmappo --out $OUT/cfg3_end_to_end_rmappo.json > $OUT/cfg3_end_to_end_rmappo.log 2>&1; tail -1 $OUT/cfg3_end_to_end_rmappo.log | cut -c1-600
