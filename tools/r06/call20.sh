#!/bin/bash
# round 6, GPU call 20: hidden-512 one-minibatch route on the resident (padded, aligned) standardised copies
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cfg_shapes.py tests/test_gpu_scripts.py tests/test_gpu_lin512.py tests/test_gpu_device_sampler_route.py tests/test_gpu_sampler_indices.py tests/test_gpu_separated.py -m gpu -q 2>&1 | tail -6 > gpurun_out/call20_tests.log
: > gpurun_out/call20.txt
for v in 1 0 1 0; do
  MAPPO_WHOLE_BATCH_VIEWS=$v timeout 600 python bench.py --workload hanabi --no-cpu-baseline --no-f32-mfma --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('views $v step', d['ms_per_step'], 'value', d['value'], 'K15 fwd', r['launch_ms'], r['frac'], 'wgrad', d['roofline_linear512_wgrad']['launch_ms'], 'peak GB', round(d['hbm_peak_bytes_per_rank'][0]/1e9,1))" >> gpurun_out/call20.txt
done
cat gpurun_out/call20.txt; tail -4 gpurun_out/call20_tests.log
