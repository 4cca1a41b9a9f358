#!/bin/bash
# round 6, GPU call 1: new tests (standardise at insert, update-graph hardening, margins), GAE in-situ probe with the
# clean / k2 modes, the driver-style bench line with `workloads`
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_margins.json
timeout 900 python -m pytest tests/test_gpu_standardize_at_insert.py tests/test_gpu_update_graph.py tests/test_gpu_device_sampler_route.py \
   tests/test_gpu_trainer_h64.py tests/test_gpu_cfg_shapes.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/call1_tests.log
echo "tests rc=$?" >> gpurun_out/call1_tests.log
timeout 300 python tools/gae_in_situ_probe.py > gpurun_out/call1_gae_probe.json 2> gpurun_out/call1_gae_probe.err
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/call1_bench.json 2> gpurun_out/call1_bench.err
echo "bench rc=$?" >> gpurun_out/call1_tests.log
tail -3 gpurun_out/call1_tests.log
