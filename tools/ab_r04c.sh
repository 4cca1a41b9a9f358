#!/bin/bash
# GPU box: (1) rollout through K9 (tests + config-3 end to end), (2) where a 64-thread SMAC shard (one rank of BASELINE config 4)
# spends its step: kernel statistics + wall clock.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_c
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_rollout_graph.py tests/test_gpu_mpe_end_to_end.py tests/test_gpu_runners.py tests/test_gpu_scripts.py -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
timeout 600 python tools/cfg3_end_to_end.py --out $OUT/cfg3_end_to_end.json > $OUT/cfg3_end_to_end.log 2>&1; tail -1 $OUT/cfg3_end_to_end.log | cut -c1-900
timeout 600 python tools/cfg3_end_to_end.py --algorithm_name rmappo --out $OUT/cfg3_end_to_end_rmappo.json > $OUT/cfg3_end_to_end_rmappo.log 2>&1; tail -1 $OUT/cfg3_end_to_end_rmappo.log | cut -c1-600
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_smac64 -o smac64 -- python $REPO/bench.py --workload smac --threads 64 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_smac64.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_cfg3roll -o cfg3roll -- python $REPO/tools/cfg3_end_to_end.py --iterations 1 > $OUT/prof_cfg3roll.log 2>&1
cd $REPO
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
tail -1 $OUT/prof_smac64.log | cut -c1-300
python - <<'PY'
import csv, os, glob
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_c/"
for name in ("prof_smac64/smac64", "prof_cfg3roll/cfg3roll"):
    f = out + name + "_kernel_stats.csv"
    if not os.path.exists(f):
        continue
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    print(name, "kernel time total %.1f ms over %d launches" % (tot / 1e6, calls))
    for r in rows[:14]:
        print("   %-80s %6s %9.1f us %5.1f%%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
