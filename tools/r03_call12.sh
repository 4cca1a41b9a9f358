#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"
OUT=gpurun_out/r03_c12; mkdir -p $OUT gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
bash tools/profile_r03.sh > $OUT/profile_r03.log 2>&1; tail -2 $OUT/profile_r03.log
timeout 900 python tools/shard_proxy.py --out gpurun_out/r03/shard_proxy.json > $OUT/shard_proxy.log 2>&1; tail -1 $OUT/shard_proxy.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
