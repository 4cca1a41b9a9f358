#!/bin/bash
# Round 5, call 5: K15 (512-wide Linear layers in six-term arithmetic) on the device -- its tests, the Hanabi-shaped bench with
# and without it, kernel statistics -- plus the update-graph tests and the whole suite after the fixes of call 4.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call5
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_lin512.py tests/test_gpu_update_graph.py "tests/test_gpu_cfg_shapes.py" -q -s -p no:cacheprovider > $OUT/new_tests.log 2>&1
echo "K15 + update-graph + cfg-shape tests rc=$?"; tail -5 $OUT/new_tests.log
for m in 1 0; do
  MAPPO_LINEAR512=$m timeout 600 python bench.py --workload hanabi --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $OUT/hanabi_k15_$m.json
done
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o hanabi -- python bench.py --workload hanabi --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-tuning > $OUT/prof.log 2>&1
s=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/r05_bench_hanabi_kernel_stats.csv
rm -rf $OUT/prof
timeout 700 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_lin512.py --deselect tests/test_gpu_update_graph.py --deselect tests/test_gpu_cfg_shapes.py > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -6 $OUT/gpu_suite.log
for g in 1 0; do
  for t in 64 128; do
    MAPPO_UPDATE_GRAPH=$g timeout 300 python bench.py --workload ns --threads $t --steps 8 --warmup 3 --no-cpu-baseline --no-f32-mfma 2>&1 | tail -1 >> $OUT/ns${t}_graph$g.jsonl
  done
done
python - <<'PY'
import json, os, glob
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/call5/"
for p in sorted(glob.glob(out + "hanabi_k15_*.json")) + sorted(glob.glob(out + "*.jsonl")):
    for l in open(p):
        if l.startswith("{"):
            r = json.loads(l)
            print(os.path.basename(p), r["ms_per_step"], r["value"], r.get("update_graph_replays_per_step"))
PY
head -8 $OUT/r05_bench_hanabi_kernel_stats.csv | cut -c1-200
grep -i "capture failed" $OUT/*.log | head -3
