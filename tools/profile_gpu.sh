#!/bin/bash
# Runs on the MI355X box (through gpurun): rocprofv3 kernel statistics of one bench step and
# HBM-traffic counters of the GAE kernel.  Results land in gpurun_out/prof_* (copy the summaries
# you want to keep into profiles/).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0

# 1. per-kernel time of the whole hot path (1 warm-up + 1 timed step, no CPU baseline)
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_bench -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
tail -2 $OUT/prof_bench.log

# 2. counters of the GAE scan (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/prof_gae_$C -o gae -- python $REPO/tools/bench_kernels.py --variants 0 --skip-gather --iters 6 --sets 6 > $OUT/prof_gae_$C.log 2>&1
done
# 3. counters of the gather
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/prof_gather_$C -o gather -- python $REPO/tools/bench_kernels.py --variants 99 --iters 2 --sets 2 --gather-variants 5 > $OUT/prof_gather_$C.log 2>&1
done
find $OUT -name "*.csv" -size +20M -delete
find $OUT -name "*.db" -delete
ls -la $OUT/prof_bench/* 2>/dev/null | head
du -sh $OUT
