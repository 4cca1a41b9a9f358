#!/bin/bash
# round 6, GPU call 7: whole-batch views on the trainer's route + the tightened tolerance tables (tests/parity.py), then the
# driver-style line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_device_sampler_route.py tests/test_gpu_trainer_h64.py tests/test_gpu_cfg_shapes.py tests/test_gpu_mid_size.py \
   tests/test_gpu_sampler_indices.py tests/test_gpu_mpe_end_to_end.py tests/test_gpu_bench.py tests/test_gpu_update_graph.py tests/test_gpu_standardize_at_insert.py -m gpu -q 2>&1 | tail -30 > gpurun_out/call7_tests.log
for v in 1 0 1 0; do
  MAPPO_WHOLE_BATCH_VIEWS=$v timeout 300 python bench.py --no-cpu-baseline --no-workloads --no-f32-mfma --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('views $v step', d['ms_per_step'], 'value', d['value'], 'gather', (d.get('roofline_gather') or {}).get('kernel'), (d.get('roofline_gather') or {}).get('launch_ms'))" >> gpurun_out/call7_views.txt
done
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/call7_bench.json 2> gpurun_out/call7_bench.err
cat gpurun_out/call7_views.txt; tail -5 gpurun_out/call7_tests.log; cut -c1-300 gpurun_out/call7_bench.json
