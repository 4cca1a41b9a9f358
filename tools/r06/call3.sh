#!/bin/bash
# round 6, GPU call 3: K15 forward without scratch (bias as the accumulators' start, two-tile MFMA groups) -- parity, then the
# Hanabi-shaped step under both group sizes (tuning bit 2 = round 5's four-tile groups), alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_lin512.py tests/test_gpu_cfg_shapes.py tests/test_gpu_standardize_at_insert.py -m gpu -q 2>&1 | tail -15 > gpurun_out/call3_tests.log
: > gpurun_out/call3_k15.txt
timeout 300 python tools/bench_lin512.py > gpurun_out/call3_lin512_two.json 2>/dev/null
MAPPO_MLP_FLAGS=2 timeout 300 python tools/bench_lin512.py > gpurun_out/call3_lin512_four.json 2>/dev/null
for f in 0 2 0 2; do
  MAPPO_MLP_FLAGS=$f timeout 600 python bench.py --workload hanabi --no-cpu-baseline --no-f32-mfma --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('flags $f step', d['ms_per_step'], 'value', d['value'], 'K15 fwd', r['launch_ms'], r['frac'], 'wgrad', d['roofline_linear512_wgrad']['launch_ms'])" >> gpurun_out/call3_k15.txt
done
cat gpurun_out/call3_k15.txt; tail -4 gpurun_out/call3_tests.log
