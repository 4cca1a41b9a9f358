#!/bin/bash
# Round 5, call 14: is test_gru_chunk_kernels_on_wide_magnitudes deterministic?  (three fresh processes, printed errors), then the
# device tests call 13 did not reach.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call14
mkdir -p $OUT
cd $REPO
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_six_term_adversarial.py -k wide_magnitudes -q -s -p no:cacheprovider 2>&1 | grep -E "K12 wide|passed|failed" >> $OUT/repeat.log
done
cat $OUT/repeat.log
timeout 1200 python -m pytest tests/test_gpu_six_term_adversarial.py tests/test_gpu_cfg_shapes.py tests/test_gpu_trainer_h64.py tests/test_gpu_bench.py tests/test_gpu_gru_seq.py -q -p no:cacheprovider --deselect tests/test_gpu_six_term_adversarial.py::test_gru_chunk_kernels_on_wide_magnitudes > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
