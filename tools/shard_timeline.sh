#!/bin/bash
# Runs on the MI355X box: kernel trace of the north-star step at one rank's share of an 8-GPU job (512 rollout threads,
# collectives through RCCL with one rank) -> where the step's fixed cost sits: device time of small kernels or idle gaps
# between launches.  python tools/shard_timeline.py <dir> summarises.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
THREADS=${1:-512}
OUT=$REPO/gpurun_out/r03/timeline_$THREADS
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
export MAPPO_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29581
timeout 300 python $REPO/bench.py --threads $THREADS --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench.json
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $REPO/bench.py --threads $THREADS --steps 4 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
find $OUT -name "*.db" -delete
cd $REPO
python tools/shard_timeline.py $OUT > $OUT/summary.json 2> $OUT/summary.err
cut -c1-400 $OUT/bench.json; head -c 3000 $OUT/summary.json
find $OUT -name "*kernel_trace.csv" -size +12M -delete
