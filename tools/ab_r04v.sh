#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the north-star step under the six-term kernels (option bits 64 + 256 + 512) and,
# for the same box, under the default kernels.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/closing
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
for f in 832 0; do
  MAPPO_MLP_FLAGS=$f timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$f -o ns -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-six-term > $OUT/prof_$f.log 2>&1
  echo "rocprof flags $f rc=$?"
  s=$(find $OUT/prof_$f -name "*kernel_stats.csv" | head -1)
  [ -n "$s" ] && cp "$s" $OUT/ns_flags${f}_kernel_stats.csv && head -12 "$s" | cut -c1-170
  rm -rf $OUT/prof_$f
done
