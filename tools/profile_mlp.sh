#!/bin/bash
# Runs on the MI355X box: per-kernel times of the fused-trunk microbenchmark (tools/bench_mlp.py) -> gpurun_out/mlp_prof/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/mlp_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
timeout 150 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o mlp -- python $REPO/tools/bench_mlp.py "$@" > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$F" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s calls %5s avg %10.1f us  total %6.2f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
P
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
