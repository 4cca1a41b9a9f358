#!/bin/bash
# First gpurun call of the next round: device runs of everything that was finished on the CPU after the GPU budget
# of round 1 was spent, then the headline bench.  Writes its logs under gpurun_out/next/.
#   gpurun --timeout 1500 -- 'bash tools/next_round_first_call.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
OUT=gpurun_out/next
mkdir -p $OUT
export MAPPO_PENDING_GPU_TESTS=1
timeout 600 python -m pytest tests/test_gpu_pending.py -q 2>&1 | tail -25 > $OUT/pending_tests.log
unset MAPPO_PENDING_GPU_TESTS
timeout 300 python -m pytest tests/test_gpu_scripts.py tests/test_gpu_runners.py -q 2>&1 | tail -8 > $OUT/script_runner_tests.log
timeout 300 python tools/device_env_check.py --threads 4096 --episodes 3 2>&1 | tail -4 > $OUT/device_env.log
timeout 120 python tools/hanabi_env_bench.py --tables 1024 --players 5 2>&1 | tail -2 > $OUT/hanabi_env.log
timeout 400 python bench.py 2>&1 | tail -1 > $OUT/bench_ns.json
tail -n 30 $OUT/*.log; cut -c1-400 $OUT/bench_ns.json
