#!/bin/bash
# K15 forward with the plane loads issued four under one M0 set-up (instruction offsets): device parity first, then the microbench.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call12
mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_lin512.py tests/test_gpu_cfg_shapes.py -q -p no:cacheprovider > $OUT/tests.log 2>&1
echo "K15 tests rc=$?"; tail -3 $OUT/tests.log
timeout 300 python tools/bench_lin512.py > $OUT/bench_lin512.json 2> $OUT/bench_lin512.err
cat $OUT/bench_lin512.json
