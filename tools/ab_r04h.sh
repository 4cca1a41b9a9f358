#!/bin/bash
# GPU box: actor and critic launch sequences of an update on two streams for small minibatches (R_MAPPOPolicy.evaluate_logits
# concurrent=True) against one stream (MAPPO_CONCURRENT_NETS=0): the whole device suite with it on, then shard-sized benches.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_h
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
MAPPO_CONCURRENT_NETS=1 timeout 900 python -m pytest tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py tests/test_gpu_parity.py -q > $OUT/gpu_tests_forced.log 2>&1; tail -2 $OUT/gpu_tests_forced.log
for i in 1 2; do
  for m in auto 0; do
    MAPPO_CONCURRENT_NETS=$m timeout 300 python bench.py --workload smac --threads 64 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/smac64_$m.jsonl
    MAPPO_CONCURRENT_NETS=$m timeout 300 python bench.py --workload ns_rnn --threads 128 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/nsrnn128_$m.jsonl
    MAPPO_CONCURRENT_NETS=$m timeout 300 python bench.py --workload cfg2 --threads 32 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/cfg2_32_$m.jsonl
  done
done
for m in 1 0; do
  MAPPO_CONCURRENT_NETS=$m timeout 300 python bench.py --workload smac --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/smac512_$m.jsonl
  MAPPO_CONCURRENT_NETS=$m timeout 300 python bench.py --workload cfg2 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/cfg2_$m.jsonl
done
python - <<'PY'
import json, os, glob
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_h/"
for f in sorted(glob.glob(out + "*.jsonl")):
    print(os.path.basename(f), [json.loads(l)["ms_per_step"] for l in open(f) if l.startswith("{")])
PY
