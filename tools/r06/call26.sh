#!/bin/bash
# round 6, GPU call 26: K9's streaming accesses as non-temporal loads / stores (build-time MAPPO_K9_NT=15) against the default
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/call26.txt
line() {
  local name=$1 lib=$2; shift; shift
  MAPPO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-f32-mfma --no-workloads "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; b=d.get('roofline_mlp_backward') or {}; print('$name step', d['ms_per_step'], 'fwd', r.get('launch_ms'), r.get('frac'), 'bwd', b.get('launch_ms'))" >> gpurun_out/call26.txt
}
D=$PWD/on-policy_amd/lib/libmappo_hip.so
N=$PWD/on-policy_amd/lib/libmappo_hip_NT15.so
for i in 1 2 3; do
  line default_ns_$i $D --steps 10 --warmup 3
  line nt15_ns_$i $N --steps 10 --warmup 3
done
line default_cfg3 $D --workload cfg3 --steps 10 --warmup 2
line nt15_cfg3 $N --workload cfg3 --steps 10 --warmup 2
line default_smac $D --workload smac --steps 10 --warmup 2
line nt15_smac $N --workload smac --steps 10 --warmup 2
MAPPO_HIP_LIB=$N timeout 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py -m gpu -q -x 2>&1 | tail -3 >> gpurun_out/call26.txt
cat gpurun_out/call26.txt
