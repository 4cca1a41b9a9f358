#!/bin/bash
# Round 5, call 13: the first-layer weight gradient inside the chain's launch (DW1; tuning bit 256 = the separate kernel) --
# device parity of the K9 tests under every fixture, then A/B on the north star, its recurrent form and the SMAC shapes, the
# kernel statistics of the fused north star, and the bench lines with the two-roof roofline objects.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call13
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_six_term_adversarial.py tests/test_gpu_cfg_shapes.py tests/test_gpu_trainer_h64.py tests/test_gpu_bench.py -q -x -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
STEPS=10 WARMUP=3 tools/ab.sh dw1_ns "ns" 2 fused:MAPPO_MLP_FLAGS=0 separate:MAPPO_MLP_FLAGS=256
STEPS=4 WARMUP=2 tools/ab.sh dw1_ns_rnn "ns_rnn" 1 fused:MAPPO_MLP_FLAGS=0 separate:MAPPO_MLP_FLAGS=256
STEPS=6 WARMUP=2 tools/ab.sh dw1_smac64 "smac --threads 64" 1 fused:MAPPO_MLP_FLAGS=0 separate:MAPPO_MLP_FLAGS=256
STEPS=6 WARMUP=2 tools/ab.sh dw1_cfg2 "cfg2" 1 fused:MAPPO_MLP_FLAGS=0 separate:MAPPO_MLP_FLAGS=256
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench_ns.json
timeout 900 python bench.py --workload hanabi --steps 2 --warmup 1 --no-cpu-baseline --no-f32-mfma 2>&1 | tail -1 > $OUT/bench_hanabi.json
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o ns -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-mfma > $OUT/prof.log 2>&1
s=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/ns_kernel_stats.csv
rm -rf $OUT/prof
python - <<'PY'
import json, os, csv
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/call13/"
for n in ("bench_ns", "bench_hanabi"):
    try:
        r = json.loads(open(out + n + ".json").read())
        print(n, r["ms_per_step"], r["value"], (r.get("f32_mfma") or {}).get("ms_per_step"), json.dumps(r["roofline"])[:600])
        print("   backward", json.dumps(r.get("roofline_mlp_backward"))[:400])
    except Exception as e:
        print(n, "unreadable", e)
try:
    rows = list(csv.DictReader(open(out + "ns_kernel_stats.csv")))
    for r in rows[:12]:
        print("%-80s %5s %9.4f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e6))
except Exception as e:
    print("no stats", e)
PY
