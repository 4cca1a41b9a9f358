#!/bin/bash
# Runs on the MI355X box (through gpurun): the round-2 evidence set -- kernel statistics of the north-star bench, HBM
# counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes) of the same command, bench lines of the other
# workloads.  Everything lands under gpurun_out/r02/ (summaries are copied into profiles/ by tools/summarize_r02.py).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_ns -o ns -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_ns.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pmc_$C -o ns -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1
done
cd $REPO
timeout 300 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 > $OUT/bench_ns.json
for w in cfg2 ns_rnn smac; do
  timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_$w.json
done
timeout 500 python bench.py --workload hanabi --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_hanabi.json
for w in cfg2 smac; do
  cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$w -o $w -- python $REPO/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_$w.log 2>&1; cd $REPO
done
timeout 120 python tools/bench_kernels.py --T 200 --N 1024 --A 5 --variants 0,70,72 --skip-gather > $OUT/kernels_cfg2.json 2> $OUT/kernels_cfg2.err
timeout 120 python tools/bench_mlp.py --reps 7 > $OUT/bench_mlp.log 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +30M -delete
ls $OUT; cut -c1-300 $OUT/bench_ns.json
