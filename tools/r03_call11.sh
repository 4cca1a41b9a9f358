#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"
OUT=gpurun_out/r03_c11; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py tests/test_gpu_sampler_indices.py tests/test_gpu_bench.py tests/test_gpu_parity.py -q -x > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee $OUT/bench_ns.json
timeout 900 python tools/shard_proxy.py --out gpurun_out/r03/shard_proxy.json > $OUT/shard_proxy.log 2>&1; tail -2 $OUT/shard_proxy.log
