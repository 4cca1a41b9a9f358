#!/bin/bash
# Round 5, call 17: K12's weight gradients in one six-term launch (mappo_gru_weight_grads) -- device parity, then A/B against the
# library GEMMs (MAPPO_GRU_WEIGHT_GRAD_KERNEL=0) on the recurrent workloads, kernel statistics of the recurrent north star.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call17
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_gru_seq.py tests/test_gpu_six_term_adversarial.py tests/test_gpu_trainer_h64.py tests/test_gpu_cfg_shapes.py tests/test_gpu_update_graph.py -q -x -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
STEPS=4 WARMUP=2 tools/ab.sh gru_wgrad_ns_rnn "ns_rnn" 2 kernel:MAPPO_GRU_WEIGHT_GRAD_KERNEL=1 library:MAPPO_GRU_WEIGHT_GRAD_KERNEL=0
STEPS=6 WARMUP=2 tools/ab.sh gru_wgrad_smac "smac" 2 kernel:MAPPO_GRU_WEIGHT_GRAD_KERNEL=1 library:MAPPO_GRU_WEIGHT_GRAD_KERNEL=0
STEPS=8 WARMUP=3 tools/ab.sh gru_wgrad_smac64 "smac --threads 64" 2 kernel:MAPPO_GRU_WEIGHT_GRAD_KERNEL=1 library:MAPPO_GRU_WEIGHT_GRAD_KERNEL=0
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o ns_rnn -- python bench.py --workload ns_rnn --steps 2 --warmup 1 --no-cpu-baseline --no-f32-mfma > $OUT/prof.log 2>&1
s=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/ns_rnn_kernel_stats.csv
rm -rf $OUT/prof
python - <<'PY'
import csv, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/call17/"
try:
    rows = list(csv.DictReader(open(out + "ns_rnn_kernel_stats.csv")))
    for r in rows[:14]:
        print("%-90s %5s %9.4f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6))
except Exception as e:
    print("no stats", e)
PY
