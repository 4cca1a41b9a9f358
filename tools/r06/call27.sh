#!/bin/bash
# round 6, GPU call 27: MAPPO_K9_NT bisected: 2 = forward's saved activations as non-temporal stores, 12 = weight-gradient DMA nt + chain's z loads nt, 14 = both
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/call27.txt
line() {
  local name=$1 lib=$2; shift; shift
  MAPPO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-f32-mfma --no-workloads "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; b=d.get('roofline_mlp_backward') or {}; print('$name step', d['ms_per_step'], 'fwd', r.get('launch_ms'), r.get('frac'), 'bwd', b.get('launch_ms'))" >> gpurun_out/call27.txt
}
L=$PWD/on-policy_amd/lib
for i in 1 2; do
  line default_ns_$i $L/libmappo_hip.so --steps 10 --warmup 3
  for v in 2 12 14; do line nt${v}_ns_$i $L/libmappo_hip_NT$v.so --steps 10 --warmup 3; done
done
for v in "" _NT12; do
  line lib${v}_smac $L/libmappo_hip$v.so --workload smac --steps 10 --warmup 2
  line lib${v}_nsrnn $L/libmappo_hip$v.so --workload ns_rnn --steps 3 --warmup 1
  line lib${v}_cfg2 $L/libmappo_hip$v.so --workload cfg2 --steps 30 --warmup 5
done
cat gpurun_out/call27.txt
