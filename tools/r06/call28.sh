#!/bin/bash
# round 6, GPU call 28: MAPPO_K9_NT=14 against the default on the small workloads (saved activations that fit the Infinity Cache)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/call28.txt
line() {
  local name=$1 lib=$2; shift; shift
  MAPPO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-f32-mfma --no-workloads "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; b=d.get('roofline_mlp_backward') or {}; print('$name step', d['ms_per_step'], 'fwd', r.get('launch_ms'), r.get('frac'), 'bwd', b.get('launch_ms'))" >> gpurun_out/call28.txt
}
L=$PWD/on-policy_amd/lib
for i in 1 2; do
for v in "" _NT14; do
  line lib${v}_shard64_$i $L/libmappo_hip$v.so --workload smac --threads 64 --steps 40 --warmup 5
  line lib${v}_cfg2_$i $L/libmappo_hip$v.so --workload cfg2 --steps 30 --warmup 5
  line lib${v}_cfg3_$i $L/libmappo_hip$v.so --workload cfg3 --steps 10 --warmup 2
  line lib${v}_ns512_$i $L/libmappo_hip$v.so --threads 512 --steps 20 --warmup 3
done
done
line lib_smac $L/libmappo_hip.so --workload smac --steps 10 --warmup 2
line lib_NT14_smac $L/libmappo_hip_NT14.so --workload smac --steps 10 --warmup 2
line lib_nsrnn $L/libmappo_hip.so --workload ns_rnn --steps 3 --warmup 1
line lib_NT14_nsrnn $L/libmappo_hip_NT14.so --workload ns_rnn --steps 3 --warmup 1
cat gpurun_out/call28.txt
