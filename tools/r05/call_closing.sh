#!/bin/bash
# Round 5, the last call on HEAD: the whole device suite and the driver's command line.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/closing
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -2 $OUT/gpu_suite.log
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench_ns.json
cut -c1-420 $OUT/bench_ns.json
