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
export TMPDIR=/tmp
MAPPO_MLP_FLAGS=832 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_six -o six -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-six-term > $OUT/prof_six.log 2>&1
echo "rocprof rc=$?"; find $OUT/prof_six -name "*kernel_stats.csv" | head -2
f=$(find $OUT/prof_six -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-160 && cp "$f" $OUT/six_term_kernel_stats.csv
find $OUT/prof_six -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
