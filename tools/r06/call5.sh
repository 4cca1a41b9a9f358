#!/bin/bash
# round 6, GPU call 5: the side stream at every size -- A/B on the large workloads, mid-size reference fixtures on the device
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mid_size.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/call5_tests.log
: > gpurun_out/call5_two_stream.txt
line() {  # name, env, bench args
  local name=$1 envs=$2; shift 2
  env $envs timeout 600 python bench.py --no-cpu-baseline --no-f32-mfma --no-workloads "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name [$envs]', 'step', d['ms_per_step'], 'value', d['value'], 'peak GB', round(d['hbm_peak_bytes_per_rank'][0]/1e9,1))" >> gpurun_out/call5_two_stream.txt
}
for rep in 1 2; do
  for ts in 0 1; do
    line ns "MAPPO_TWO_STREAM_UPDATE=$ts" --workload ns --steps 10 --warmup 3
    line ns_rnn "MAPPO_TWO_STREAM_UPDATE=$ts" --workload ns_rnn --steps 4 --warmup 2
    line cfg3 "MAPPO_TWO_STREAM_UPDATE=$ts" --workload cfg3 --steps 8 --warmup 2
    line ns_rnn_shard128 "MAPPO_TWO_STREAM_UPDATE=$ts" --workload ns_rnn --threads 128 --steps 10 --warmup 3
  done
done
for ts in 0 1; do
  line hanabi "MAPPO_TWO_STREAM_UPDATE=$ts" --workload hanabi --steps 2 --warmup 1
done
cat gpurun_out/call5_two_stream.txt; tail -6 gpurun_out/call5_tests.log
