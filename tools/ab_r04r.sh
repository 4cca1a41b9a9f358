#!/bin/bash
# GPU box: the six-term bf16 forms (option bits 64 = version-4 forward for widths >= 128, 256 = direct first-layer
# weight-gradient kernel) against the default float32-MFMA kernels: K9 device tests under both, reference-generated
# trainer fixtures under the flags, kernel times at the north-star shapes, the north-star step, alternating on one box.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_r
mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_mlp.py -q -p no:cacheprovider > $OUT/gpu_mlp_both.log 2>&1
echo "test_gpu_mlp (default + bf16x6 params) rc=$?"; tail -3 $OUT/gpu_mlp_both.log
for f in 0 320; do
  MAPPO_MLP_FLAGS=$f timeout 200 python tools/bench_mlp.py --sequential --din 384 48 --reps 5 > $OUT/bench_mlp_flag$f.log 2>&1
  echo "bench_mlp flag $f rc=$?"; grep "^{" $OUT/bench_mlp_flag$f.log | cut -c1-420
done
MAPPO_MLP_FLAGS=320 timeout 300 python -m pytest tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py -q -p no:cacheprovider > $OUT/gpu_trainer_flag320.log 2>&1
echo "trainer fixtures under flags 64 + 256 rc=$?"; tail -2 $OUT/gpu_trainer_flag320.log
for i in 1 2; do
  for f in 0 64 320; do
    MAPPO_MLP_FLAGS=$f timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/ns_flag$f.jsonl
  done
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_r/"
for name in ("ns_flag0", "ns_flag64", "ns_flag320"):
    rows = [json.loads(l) for l in open(out + name + ".jsonl") if l.startswith("{")]
    print(name, [r["ms_per_step"] for r in rows], [r["roofline"]["launch_ms"] for r in rows], [r["roofline_mlp_backward"]["launch_ms"] for r in rows])
PY
