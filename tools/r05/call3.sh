#!/bin/bash
# Round 5, call 3: ppo_update as a captured HIP graph -- its own tests, the whole device suite with it on (the default), and
# the A / B graph on / off on the launch-bound workloads.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call3
mkdir -p $OUT
cd $REPO
timeout 400 python -m pytest tests/test_gpu_update_graph.py tests/test_gpu_six_term_adversarial.py -q -x -p no:cacheprovider > $OUT/graph_tests.log 2>&1
echo "update-graph + adversarial tests rc=$?"; tail -4 $OUT/graph_tests.log
timeout 700 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_update_graph.py --deselect tests/test_gpu_six_term_adversarial.py > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -6 $OUT/gpu_suite.log
run() { # name graph workload extra...
  name=$1; g=$2; shift 2
  MAPPO_UPDATE_GRAPH=$g timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32-mfma "$@" 2>&1 | tail -1 >> $OUT/${name}_graph$g.jsonl
}
for rep in 1 2; do
  for g in 1 0; do
    run cfg2 $g --workload cfg2
    run smac64 $g --workload smac --threads 64
    run smac $g --workload smac
    run ns512 $g --workload ns --threads 512
  done
done
for g in 1 0; do run cfg3 $g --workload cfg3; run ns_rnn128 $g --workload ns_rnn --threads 128; done
MAPPO_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29591 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32-mfma --workload smac --threads 64 2>&1 | tail -1 > $OUT/smac64_one_rank_rccl_graph1.jsonl
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o s64 -- python bench.py --workload smac --threads 64 --steps 8 --warmup 2 --no-cpu-baseline --no-f32-mfma > $OUT/prof.log 2>&1
s=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/r05_smac64_kernel_stats.csv
rm -rf $OUT/prof
python - <<'PY'
import json, os, glob
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/call3/"
for p in sorted(glob.glob(out + "*.jsonl")):
    rows = [json.loads(l) for l in open(p) if l.startswith("{")]
    print(os.path.basename(p), [r["ms_per_step"] for r in rows], [r.get("update_graph_replays_per_step") for r in rows])
PY
grep -i "capture failed" $OUT/*.log | head
