#!/bin/bash
# round 6, GPU call 15: rooflines of two-stream workloads from the extra one-stream step; is config 3's step bimodal?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/call15.txt
line() {
  local name=$1; shift
  timeout 600 python bench.py --no-cpu-baseline --no-f32-mfma --no-workloads "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name step', d['ms_per_step'], 'two_streams', d['two_streams'], 'roof', r['frac'], r['launch_ms'], r['timed_in'][:60], 'peak GB', round(d['hbm_peak_bytes_per_rank'][0]/1e9,1))" >> gpurun_out/call15.txt
}
line hanabi --workload hanabi --steps 3 --warmup 1
line smac --workload smac --steps 10 --warmup 2
line cfg2 --workload cfg2 --steps 30 --warmup 5
for i in 1 2 3 4 5 6; do line cfg3_$i --workload cfg3 --steps 10 --warmup 2; done
for i in 1 2 3; do line cfg3_20steps_$i --workload cfg3 --steps 20 --warmup 5; done
cat gpurun_out/call15.txt
