#!/bin/bash
# round 6, GPU call 25: GAE timed by the kernel's own begin / end timestamps (mappo_gae_time_next_launch) next to the event pair
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "gae_dispatch" 2>&1 | tail -25 > gpurun_out/call25_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-workloads --no-f32-mfma --steps 10 --warmup 3 2>gpurun_out/call25_bench.err | tail -1 > gpurun_out/call25_bench.json
python -c "
import json; d=json.load(open('gpurun_out/call25_bench.json')); print(d['ms_per_step'], json.dumps(d['roofline_gae'])[:1500])"
timeout 600 python bench.py --workload cfg3 --no-cpu-baseline --no-f32-mfma --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', d['ms_per_step'], json.dumps(d['roofline_gae'])[:600])"
tail -8 gpurun_out/call25_tests.log
