#!/bin/bash
# Round 5, closing call on HEAD: the whole device suite, the driver's line, the Hanabi-shaped line (K15's launches in its roofline
# objects) with its kernel statistics, the 64-thread SMAC shard through one-rank RCCL.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05/final
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -3 $OUT/gpu_suite.log
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench_ns.json
timeout 900 python bench.py --workload hanabi --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_hanabi.json
MAPPO_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29593 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --workload smac --threads 64 --steps 8 --warmup 3 --no-cpu-baseline --no-f32-mfma 2>&1 | tail -1 > $OUT/bench_smac64_one_rank_rccl.json
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o hanabi -- python bench.py --workload hanabi --steps 1 --warmup 1 --no-cpu-baseline --no-f32-mfma --no-gemm-tuning > $OUT/prof.log 2>&1
s=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && cp "$s" $OUT/r05_bench_hanabi_kernel_stats.csv
rm -rf $OUT/prof
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/final/"
for n in ("bench_ns", "bench_hanabi", "bench_smac64_one_rank_rccl"):
    try:
        r = json.loads(open(out + n + ".json").read())
        print(n, r["ms_per_step"], r["value"], r.get("f32_mfma", {}).get("ms_per_step"), r["roofline"]["kernel"][:30], r["roofline"]["frac"], r.get("scalar_allreduce"), r.get("arithmetic", "")[:40])
    except Exception as e:
        print(n, "unreadable", e)
PY
