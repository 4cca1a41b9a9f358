#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$REPO"
OUT=gpurun_out/r03_c7; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gru_seq.py tests/test_gpu_trainer_h64.py tests/test_gpu_parity.py -q -k "gru or h64 or train_on_device" > $OUT/tests.log 2>&1
tail -6 $OUT/tests.log
for wl in smac ns_rnn; do
  echo "--- $wl chunk kernel"; timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-300 | tee $OUT/bench_$wl.json
  echo "--- $wl step kernels"; MAPPO_GRU_CHUNK=0 timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-300 | tee $OUT/bench_${wl}_steps.json
done
bash tools/profile_bench.sh ns_rnn --workload ns_rnn > $OUT/prof_ns_rnn.txt 2>&1; cat $OUT/prof_ns_rnn.txt
