#!/bin/bash
# GPU box, last call of round 4: (1) the bf16-split probe (tools/probes/probe_bf16_split.hip: accuracy of float32 products
# formed from 3 / 6 / 9 bf16 x bf16 terms on the matrix core, and what a k = 16 step costs a SIMD that way) -- the data
# the next round's decision about K9's arithmetic needs; (2) the whole device suite on HEAD; (3) the driver's bench line.
#   gpurun --timeout 840 -- 'bash tools/ab_r04o.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/final
mkdir -p $OUT
cd $REPO
timeout 120 tools/probes/probe_bf16_split > $OUT/probe_bf16_split.jsonl 2> $OUT/probe_bf16_split.err
echo "probe rc=$?"; cat $OUT/probe_bf16_split.jsonl
timeout 420 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -3 $OUT/gpu_suite.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_ns.log 2>&1
echo "bench rc=$?"; tail -1 $OUT/bench_ns.log | cut -c1-600
