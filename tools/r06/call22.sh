#!/bin/bash
# round 6, GPU call 22: K15 forward with the block epilogue (bias + ReLU + LayerNorm): device tests + Hanabi-shaped step A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lin512.py tests/test_gpu_cfg_shapes.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/call22_tests.log
: > gpurun_out/call22.txt
for v in 1 0 1 0; do
  MAPPO_LINEAR512_NORM=$v timeout 600 python bench.py --workload hanabi --no-cpu-baseline --no-f32-mfma --steps 3 --warmup 1 2>gpurun_out/call22_bench_$v.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('norm-epilogue $v step', d['ms_per_step'], 'value', d['value'], 'K15 fwd', r['launch_ms'], r['frac'], 'wgrad', d['roofline_linear512_wgrad']['launch_ms'], 'peak GB', round(d['hbm_peak_bytes_per_rank'][0]/1e9,1))" >> gpurun_out/call22.txt
done
cat gpurun_out/call22.txt; tail -12 gpurun_out/call22_tests.log
