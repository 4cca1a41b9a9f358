#!/bin/bash
# GPU box: cycle stamps of the version-4 forward (wave 0 of workgroup 0: chunk loop / tails per tile) at the critic and actor
# shapes, next to version 3's.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_q
mkdir -p $OUT
cd $REPO
for f in 64 0; do
  MAPPO_MLP_FLAGS=$f timeout 200 python tools/bench_mlp.py --sequential --din 384 48 --reps 3 --stamps > $OUT/stamps_flag$f.log 2>&1
  echo "flag $f rc=$?"; grep -v amdgpu.ids $OUT/stamps_flag$f.log | cut -c1-200 | head -40
done
