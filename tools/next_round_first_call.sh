#!/bin/bash
# First gpurun call of the next round: the whole device suite, the round-3 evidence set again on the current code and the
# shard proxy.  Logs under gpurun_out/next/.
#   gpurun --timeout 2400 -- 'bash tools/next_round_first_call.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/next
mkdir -p $OUT gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
bash tools/profile_r03.sh > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
timeout 900 python tools/shard_proxy.py --out gpurun_out/r03/shard_proxy.json > $OUT/shard_proxy.log 2>&1; tail -1 $OUT/shard_proxy.log
