#!/bin/bash
# GPU box, closing call of round 4: the whole device suite on HEAD, the driver's bench line (default kernels; it now also
# carries the opt-in six-term figure), rocprofv3 kernel statistics of the north-star step under the six-term kernels.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/closing
mkdir -p $OUT
cd $REPO
timeout 420 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -3 $OUT/gpu_suite.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_ns.log 2>&1
echo "bench rc=$?"; tail -1 $OUT/bench_ns.log | cut -c1-300
timeout 200 python tools/six_term_accuracy.py > $OUT/six_term_accuracy.json 2> $OUT/six_term_accuracy.err
echo "accuracy rc=$?"; python - <<'PY'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/closing/six_term_accuracy.json"))
for r in d["rows"]:
    print(r["din"], r["kernels"], "y %.2e" % r["max_error_over_largest_entry_vs_float64"]["y"], "worst %.2e" % r["worst"])
PY
for f in 0 960; do
  MAPPO_MLP_FLAGS=$f timeout 300 python -m pytest tests/test_gpu_trainer_h64.py -q -s -k fused_trunk_update -p no:cacheprovider 2>&1 | grep "largest relative errors" > $OUT/trainer_fixture_errors_flag$f.txt
  echo "fixture errors flag $f: $(wc -l < $OUT/trainer_fixture_errors_flag$f.txt) cases"
done
export TMPDIR=/tmp
MAPPO_MLP_FLAGS=832 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_six -o six -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-six-term > $OUT/prof_six.log 2>&1
echo "rocprof rc=$?"; find $OUT/prof_six -name "*kernel_stats.csv" | head -2
f=$(find $OUT/prof_six -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-160 && cp "$f" $OUT/six_term_kernel_stats.csv
find $OUT/prof_six -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
