#!/bin/bash
# Round 5, call 15: why does test_gru_chunk_kernels_on_wide_magnitudes pass inside the whole suite and fail alone?  The whole suite
# with its printed errors, then the test behind three groups of the modules that precede it.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call15
mkdir -p $OUT
cd $REPO
W=tests/test_gpu_six_term_adversarial.py::test_gru_chunk_kernels_on_wide_magnitudes
timeout 1200 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $OUT/suite.log 2>&1
echo "suite rc=$?"; grep -E "K12 wide|passed|failed" $OUT/suite.log | cut -c1-900
run() { name=$1; shift; timeout 600 python -m pytest "$@" $W -q -m gpu -s -p no:cacheprovider > $OUT/$name.log 2>&1; echo "$name rc=$?"; grep -E "K12 wide|passed|failed" $OUT/$name.log | cut -c1-420; }
run a tests/test_gpu_action_spaces.py tests/test_gpu_fused_loss.py tests/test_gpu_gru_seq.py tests/test_gpu_lin512.py
run b tests/test_gpu_optim.py tests/test_gpu_parity.py tests/test_gpu_pending.py tests/test_gpu_rollout_graph.py tests/test_gpu_mat.py
run c tests/test_gpu_runners.py tests/test_gpu_sampler_indices.py tests/test_gpu_scripts.py tests/test_gpu_separated.py tests/test_gpu_mpe_end_to_end.py tests/test_gpu_device_sampler_route.py
