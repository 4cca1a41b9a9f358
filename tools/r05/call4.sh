#!/bin/bash
# Round 5, call 4: the update graph on the feed-forward route as well (stride-0 RNN-state views are read in place): its tests,
# the whole device suite, A / B graph on / off on the feed-forward workloads, kernel statistics of the 64-thread SMAC shard.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/call4
mkdir -p $OUT
cd $REPO
timeout 400 python -m pytest tests/test_gpu_update_graph.py -q -p no:cacheprovider > $OUT/graph_tests.log 2>&1
echo "update-graph tests rc=$?"; tail -4 $OUT/graph_tests.log
timeout 700 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_update_graph.py > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -6 $OUT/gpu_suite.log
run() { # name graph workload extra...
  name=$1; g=$2; shift 2
  MAPPO_UPDATE_GRAPH=$g timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32-mfma "$@" 2>&1 | tail -1 >> $OUT/${name}_graph$g.jsonl
}
for rep in 1 2; do
  for g in 1 0; do
    run cfg2 $g --workload cfg2
    run ns512 $g --workload ns --threads 512
    run cfg3 $g --workload cfg3
  done
done
for g in 1 0; do run smac64 $g --workload smac --threads 64; run ns $g --workload ns; done
MAPPO_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29591 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-f32-mfma --workload ns --threads 512 2>&1 | tail -1 > $OUT/ns512_one_rank_rccl_graph1.jsonl
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o s64 -- python bench.py --workload smac --threads 64 --steps 8 --warmup 2 --no-cpu-baseline --no-f32-mfma > $OUT/prof.log 2>&1
s=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/r05_smac64_kernel_stats.csv
rm -rf $OUT/prof
python - <<'PY'
import json, os, glob
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/call4/"
for p in sorted(glob.glob(out + "*.jsonl")):
    rows = [json.loads(l) for l in open(p) if l.startswith("{")]
    print(os.path.basename(p), [r["ms_per_step"] for r in rows], [r.get("update_graph_replays_per_step") for r in rows])
PY
grep -i "capture failed" $OUT/*.log | head
