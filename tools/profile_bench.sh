#!/bin/bash
# Runs on the MI355X box: rocprofv3 kernel statistics of one bench.py workload -> gpurun_out/bench_prof/<name>_kernel_stats.csv
#   tools/profile_bench.sh [name] [bench.py arguments ...]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=${1:-ns}; shift
OUT=$REPO/gpurun_out/bench_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o $NAME -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/$NAME.log 2>&1
F=$(find $OUT -name "${NAME}_kernel_stats.csv" | head -1)
python - "$F" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-64s calls %5s avg %9.1f us  %6.2f%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
P
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
