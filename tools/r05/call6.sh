#!/bin/bash
# Round 5, call 6: K15 with the MFMAs of a step interleaved over the tiles -- tests, Hanabi-shaped bench, kernel statistics.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call6
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_lin512.py tests/test_gpu_update_graph.py tests/test_gpu_cfg_shapes.py tests/test_gpu_parity.py -q -p no:cacheprovider > $OUT/new_tests.log 2>&1
echo "K15 + update-graph + cfg-shape + parity tests rc=$?"; tail -5 $OUT/new_tests.log
timeout 600 python bench.py --workload hanabi --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $OUT/hanabi_k15_1.json
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o hanabi -- python bench.py --workload hanabi --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-tuning > $OUT/prof.log 2>&1
s=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/r05_bench_hanabi_kernel_stats.csv
rm -rf $OUT/prof
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/call6/"
r = json.loads(open(out + "hanabi_k15_1.json").read()); print("hanabi", r["ms_per_step"], r["value"], r["hbm_peak_bytes_per_rank"])
PY
head -12 $OUT/r05_bench_hanabi_kernel_stats.csv | cut -c1-150
