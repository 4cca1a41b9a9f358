#!/bin/bash
# GPU box: version 4 of the K9 forward (option bit 64: first layer as six bf16 x bf16 terms per float32 product) against
# version 3 -- device parity of the K9 tests and of the reference-generated trainer fixtures under the flag, kernel times
# at the north-star shapes, the north-star step, alternating on one box.
#   gpurun --timeout 840 -- 'bash tools/ab_r04p.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_p
mkdir -p $OUT
cd $REPO
MAPPO_MLP_FLAGS=64 timeout 300 python -m pytest tests/test_gpu_mlp.py -q -p no:cacheprovider > $OUT/gpu_mlp_flag64.log 2>&1
echo "test_gpu_mlp under flag 64 rc=$?"; tail -4 $OUT/gpu_mlp_flag64.log
for f in 0 64; do
  MAPPO_MLP_FLAGS=$f timeout 200 python tools/bench_mlp.py --sequential --din 384 48 --reps 5 > $OUT/bench_mlp_flag$f.log 2>&1
  echo "bench_mlp flag $f rc=$?"; tail -4 $OUT/bench_mlp_flag$f.log | cut -c1-400
done
MAPPO_MLP_FLAGS=64 timeout 300 python -m pytest tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py -q -p no:cacheprovider > $OUT/gpu_trainer_flag64.log 2>&1
echo "trainer fixtures under flag 64 rc=$?"; tail -4 $OUT/gpu_trainer_flag64.log
for i in 1 2; do
  for f in 0 64; do
    MAPPO_MLP_FLAGS=$f timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/ns_flag$f.jsonl
  done
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_p/"
for name in ("ns_flag0", "ns_flag64"):
    rows = [json.loads(l) for l in open(out + name + ".jsonl") if l.startswith("{")]
    print(name, [r["ms_per_step"] for r in rows], [r["roofline"]["launch_ms"] for r in rows])
PY
