#!/bin/bash
# round 6, GPU call 2: the new tests again, then the GAE ring-depth variants in situ (bench.py's own measurement)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_margins.json
timeout 1200 python -m pytest tests/test_gpu_standardize_at_insert.py tests/test_gpu_update_graph.py tests/test_gpu_device_sampler_route.py \
   tests/test_gpu_trainer_h64.py tests/test_gpu_cfg_shapes.py -m gpu -q 2>&1 | tail -40 > gpurun_out/call2_tests.log
: > gpurun_out/call2_gae_variants.txt
for v in 3057 3058 3059 3060 3061 3062 3057 3059; do
  MAPPO_GAE_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --no-workloads --no-f32-mfma --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); g=d['roofline_gae']; print('variant $v in_situ', g['in_situ'], 'b2b', g['back_to_back']['frac'], g['back_to_back']['launch_ms'], 'step', d['ms_per_step'], 'kernel variant', g['variant'])" >> gpurun_out/call2_gae_variants.txt
done
cat gpurun_out/call2_gae_variants.txt; tail -5 gpurun_out/call2_tests.log
