#!/bin/bash
# GPU box: + the six-term form of the backward chain's two 64 x 64 products (option bit 512) -- K9 device tests under the
# default kernels and under all six-term forms, the reference-generated trainer fixtures under the flags, kernel times,
# the north-star / config-2 / config-3 steps alternating on one box.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_s
mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_mlp.py -q -p no:cacheprovider > $OUT/gpu_mlp_both.log 2>&1
echo "test_gpu_mlp (default + bf16x6 params) rc=$?"; tail -3 $OUT/gpu_mlp_both.log
for f in 0 832; do
  MAPPO_MLP_FLAGS=$f timeout 200 python tools/bench_mlp.py --sequential --din 384 48 --reps 5 > $OUT/bench_mlp_flag$f.log 2>&1
  echo "bench_mlp flag $f rc=$?"; grep "^{" $OUT/bench_mlp_flag$f.log | cut -c1-420
done
MAPPO_MLP_FLAGS=832 timeout 300 python -m pytest tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py -q -p no:cacheprovider > $OUT/gpu_trainer_flag832.log 2>&1
echo "trainer fixtures under flags 64 + 256 + 512 rc=$?"; tail -2 $OUT/gpu_trainer_flag832.log
for i in 1 2; do
  for f in 0 320 832; do
    MAPPO_MLP_FLAGS=$f timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-six-term 2>&1 | tail -1 >> $OUT/ns_flag$f.jsonl
  done
done
for w in cfg2 cfg3; do
  for f in 0 832; do
    MAPPO_MLP_FLAGS=$f timeout 200 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-six-term 2>&1 | tail -1 >> $OUT/${w}_flag$f.jsonl
  done
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_s/"
for name in ("ns_flag0", "ns_flag320", "ns_flag832", "cfg2_flag0", "cfg2_flag832", "cfg3_flag0", "cfg3_flag832"):
    rows = [json.loads(l) for l in open(out + name + ".jsonl") if l.startswith("{")]
    print(name, [r["ms_per_step"] for r in rows], [r["roofline"]["launch_ms"] for r in rows], [r["roofline_mlp_backward"]["launch_ms"] for r in rows])
PY
