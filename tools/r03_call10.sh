#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"
OUT=gpurun_out/r03_c10; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py -q > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 400 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee $OUT/bench_ns.json
bash tools/profile_bench.sh ns > $OUT/prof_ns.txt 2>&1; head -8 $OUT/prof_ns.txt
MAPPO_FORCE_DIST=1 bash tools/profile_bench.sh ns512 --threads 512 > $OUT/prof_ns512.txt 2>&1; head -30 $OUT/prof_ns512.txt
