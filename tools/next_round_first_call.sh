#!/bin/bash
# First gpurun call of the next round: the whole device suite (default kernels, then with every opt-in six-term bit set for
# the process), the bench lines of every workload under both arithmetic forms, the accuracy tool, kernel statistics.
#   gpurun --timeout 1500 -- 'bash tools/next_round_first_call.sh'
# If VERDICT accepts the six-term arithmetic as the reference's float32: flip the default in csrc/mappo_mlp_impl.h
# (tuning_flags_ref: `v = e ? atoi(e) : 0` -> 64 + 256 + 512 + 1024), keep bit 4096 (to be added) as the way back to the
# float32-MFMA kernels, and move bench.py's secondary measurement to the float32-MFMA kernels.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/first
mkdir -p $OUT
cd $REPO
timeout 420 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -2 $OUT/gpu_suite.log
MAPPO_MLP_FLAGS=1856 timeout 420 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/gpu_suite_flags1856.log 2>&1
echo "suite under flags 1856 rc=$?"; tail -2 $OUT/gpu_suite_flags1856.log
for w in ns cfg2 cfg3 ns_rnn smac; do
  for f in 0 1856; do
    MAPPO_MLP_FLAGS=$f timeout 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-six-term 2>&1 | tail -1 >> $OUT/${w}_flag$f.jsonl
  done
done
# the three forms written at the end of round 4 on the emulator only (bits 2048, 4096, 8192): device parity first, then their A / B
MAPPO_MLP_FLAGS=16192 MAPPO_TEST_EXTRA_FLAGS=14336 timeout 300 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_gru_seq.py tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py -q -p no:cacheprovider > $OUT/gpu_pending_bits.log 2>&1
echo "K9 / K12 tests + fixtures with the process-wide flags 16192 rc=$?"; tail -2 $OUT/gpu_pending_bits.log
for w in ns ns_rnn smac; do
  for f in 1856 16192; do
    MAPPO_MLP_FLAGS=$f timeout 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-six-term 2>&1 | tail -1 >> $OUT/pending_${w}_flag$f.jsonl
  done
done
timeout 200 python tools/six_term_accuracy.py > $OUT/six_term_accuracy.json 2> $OUT/six_term_accuracy.err
export TMPDIR=/tmp
for f in 0 1856; do
  MAPPO_MLP_FLAGS=$f timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$f -o ns_rnn -- python bench.py --workload ns_rnn --steps 2 --warmup 1 --no-cpu-baseline --no-six-term > $OUT/prof_$f.log 2>&1
  s=$(find $OUT/prof_$f -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/ns_rnn_flags${f}_kernel_stats.csv
  rm -rf $OUT/prof_$f
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/first/"
for w in ("ns", "cfg2", "cfg3", "ns_rnn", "smac"):
    for f in (0, 1856):
        rows = [json.loads(l) for l in open(out + "%s_flag%d.jsonl" % (w, f)) if l.startswith("{")]
        print(w, f, [r["ms_per_step"] for r in rows], [r["value"] for r in rows])
for w in ("ns", "ns_rnn", "smac"):
    for f in (1856, 16192):
        rows = [json.loads(l) for l in open(out + "pending_%s_flag%d.jsonl" % (w, f)) if l.startswith("{")]
        print("pending bits", w, f, [r["ms_per_step"] for r in rows])
PY
