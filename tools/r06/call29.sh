#!/bin/bash
# round 6, GPU call 29: MAPPO_K9_NT=30 (14 + the chain's dz1 rows as non-temporal stores) against 14 and the default
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/call29.txt
line() {
  local name=$1 lib=$2; shift; shift
  MAPPO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-f32-mfma --no-workloads "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; b=d.get('roofline_mlp_backward') or {}; print('$name step', d['ms_per_step'], 'fwd', r.get('launch_ms'), r.get('frac'), 'bwd', b.get('launch_ms'))" >> gpurun_out/call29.txt
}
L=$PWD/on-policy_amd/lib
for i in 1 2 3; do
for v in "" _NT14 _NT30; do
  line lib${v}_ns_$i $L/libmappo_hip$v.so --steps 10 --warmup 3
done
done
for v in _NT14 _NT30; do
  line lib${v}_cfg3 $L/libmappo_hip$v.so --workload cfg3 --steps 10 --warmup 2
  line lib${v}_smac $L/libmappo_hip$v.so --workload smac --steps 10 --warmup 2
done
cat gpurun_out/call29.txt
