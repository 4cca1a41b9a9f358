#!/bin/bash
# round 3, first GPU call: hidden-64 reference parity of the fused route, headline bench, rocprof of the recurrent workloads
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"
OUT=gpurun_out/r03_c1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_trainer_h64.py -q -s > $OUT/h64_tests.log 2>&1
timeout 400 python bench.py 2>&1 | tail -1 > $OUT/bench_ns.json
timeout 400 python bench.py --no-cpu-baseline --sampler-rng host --steps 5 --warmup 2 2>&1 | tail -1 > $OUT/bench_ns_host.json
bash tools/profile_bench.sh ns_rnn --workload ns_rnn > $OUT/prof_ns_rnn.txt 2>&1
bash tools/profile_bench.sh smac --workload smac > $OUT/prof_smac.txt 2>&1
cat $OUT/h64_tests.log; cut -c1-600 $OUT/bench_ns.json; echo; cut -c1-300 $OUT/bench_ns_host.json; echo; cat $OUT/prof_ns_rnn.txt $OUT/prof_smac.txt
