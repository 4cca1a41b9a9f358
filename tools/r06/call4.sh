#!/bin/bash
# round 6, GPU call 4: the critic on a side stream (R_MAPPOPolicy.evaluate_logits) -- parity, then A/B on the small shards
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_update_graph.py tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py tests/test_gpu_bench.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/call4_tests.log
: > gpurun_out/call4_two_stream.txt
line() {  # name, env, bench args
  local name=$1 envs=$2; shift 2
  env $envs timeout 600 python bench.py --no-cpu-baseline --no-f32-mfma --no-workloads "$@" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name [$envs]', 'step', d['ms_per_step'], 'value', d['value'], 'replays', d['update_graph_replays_per_step'])" >> gpurun_out/call4_two_stream.txt
}
for rep in 1 2; do
  for ts in 0 1; do
    line smac_shard64 "MAPPO_TWO_STREAM_UPDATE=$ts" --workload smac --threads 64 --steps 10 --warmup 3
    line ns_rnn_shard128 "MAPPO_TWO_STREAM_UPDATE=$ts" --workload ns_rnn --threads 128 --steps 10 --warmup 3
    line ns_shard64 "MAPPO_TWO_STREAM_UPDATE=$ts" --workload ns --threads 64 --steps 10 --warmup 3
    line cfg2 "MAPPO_TWO_STREAM_UPDATE=$ts MAPPO_TWO_STREAM_MAX_ROWS=2000000" --workload cfg2 --steps 10 --warmup 3
    line smac "MAPPO_TWO_STREAM_UPDATE=$ts MAPPO_TWO_STREAM_MAX_ROWS=2000000" --workload smac --steps 5 --warmup 2
  done
done
line ns_two_stream "MAPPO_TWO_STREAM_UPDATE=1 MAPPO_TWO_STREAM_MAX_ROWS=100000000" --workload ns --steps 8 --warmup 2
line ns_one_stream "MAPPO_TWO_STREAM_UPDATE=0" --workload ns --steps 8 --warmup 2
cat gpurun_out/call4_two_stream.txt; tail -4 gpurun_out/call4_tests.log
