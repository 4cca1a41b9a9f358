#!/bin/bash
# round 6, GPU call 21: GAE in situ, option bits of the default variant (bit 0 XCD-contiguous strips, bit 1 non-temporal DMA) and
# the other strip shapes once more on the final code; 12 timed steps per line, two passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/call21.txt
for pass in 1 2; do
for v in 3057 57 1057 2057 3051 3054 3042; do
  MAPPO_GAE_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --no-workloads --no-f32-mfma --steps 12 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); g=d['roofline_gae']; print('variant $v in_situ', g['in_situ']['launch_ms'], g['in_situ']['frac'], 'b2b', g['back_to_back']['frac'], 'kernel', g['variant'])" >> gpurun_out/call21.txt
done
done
cat gpurun_out/call21.txt
