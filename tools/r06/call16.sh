#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/call16.txt
line() {
  local name=$1; shift
  timeout 600 python bench.py --no-cpu-baseline --no-f32-mfma --no-workloads "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name step', d['ms_per_step'])" >> gpurun_out/call16.txt
}
for i in 1 2 3 4 5 6 7 8; do line cfg3_$i --workload cfg3 --steps 10 --warmup 2; done
for i in 1 2 3; do line cfg2_$i --workload cfg2 --steps 30 --warmup 5; done
cat gpurun_out/call16.txt
