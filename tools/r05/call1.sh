#!/bin/bash
# Round 5, call 1: device parity of the three forms that had only run on the emulator (option bits 2048 / 4096 / 8192),
# then their A / B per bit against the verified six-term set (1856 = 64 + 256 + 512 + 1024).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call1
mkdir -p $OUT
cd $REPO
MAPPO_MLP_FLAGS=16192 MAPPO_TEST_EXTRA_FLAGS=14336 timeout 400 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_gru_seq.py tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py -q -p no:cacheprovider > $OUT/gpu_pending_bits.log 2>&1
echo "K9 / K12 tests + fixtures with the process-wide flags 16192 rc=$?"; tail -3 $OUT/gpu_pending_bits.log
run() { # workload flags
  MAPPO_MLP_FLAGS=$2 timeout 200 python bench.py --workload $1 --steps 6 --warmup 2 --no-cpu-baseline --no-six-term 2>&1 | tail -1 >> $OUT/$1_flag$2.jsonl
}
for rep in 1 2; do
  for f in 1856 3904 5952 8000; do run ns $f; done
done
for rep in 1 2; do
  for f in 1856 10048 16192; do run smac $f; done
done
for f in 1856 10048 16192; do run ns_rnn $f; done
python - <<'PY'
import json, os, glob
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/call1/"
for p in sorted(glob.glob(out + "*.jsonl")):
    rows = [json.loads(l) for l in open(p) if l.startswith("{")]
    print(os.path.basename(p), [r["ms_per_step"] for r in rows])
PY
