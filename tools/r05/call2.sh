#!/bin/bash
# Round 5, call 2: the whole device suite on the new default (six-term arithmetic as a per-call field), the adversarial and
# cfg-shape tests, then bench lines of every workload and a kernel-stats pass of the north star.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call2
mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_six_term_adversarial.py tests/test_gpu_cfg_shapes.py -q -s -p no:cacheprovider > $OUT/new_tests.log 2>&1
echo "adversarial + cfg-shape tests rc=$?"; tail -3 $OUT/new_tests.log
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x --deselect tests/test_gpu_six_term_adversarial.py --deselect tests/test_gpu_cfg_shapes.py > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -3 $OUT/gpu_suite.log
timeout 200 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench_ns_driver_line.json
for w in cfg2 cfg3 ns_rnn smac; do
  timeout 300 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_$w.json
done
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o ns -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-f32-mfma > $OUT/prof.log 2>&1
s=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/r05_bench_ns_kernel_stats.csv
rm -rf $OUT/prof
python - <<'PY'
import json, os, glob
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/call2/"
for p in sorted(glob.glob(out + "bench_*.json")):
    try:
        r = json.loads(open(p).read())
        print(os.path.basename(p), r["ms_per_step"], r["value"], r.get("f32_mfma", {}).get("ms_per_step"), r["roofline"]["frac"], r.get("hbm_peak_bytes_per_rank"))
    except Exception as e:
        print(p, "unreadable", e)
PY
head -12 $OUT/r05_bench_ns_kernel_stats.csv
