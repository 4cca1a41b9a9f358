#!/bin/bash
# GPU box: the WHOLE device suite with the six-term kernels switched on for the process (MAPPO_MLP_FLAGS=832): runners, train
# scripts, rollout graph, data-parallel tests, reference fixtures -- everything that reaches K9 takes the opt-in kernels.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/closing
mkdir -p $OUT
cd $REPO
MAPPO_MLP_FLAGS=832 timeout 500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/gpu_suite_flags832.log 2>&1
echo "suite under flags 832 rc=$?"; tail -4 $OUT/gpu_suite_flags832.log
