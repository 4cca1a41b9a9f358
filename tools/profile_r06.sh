#!/bin/bash
# Runs on the MI355X box (through gpurun): the round-6 evidence set.  Stages (first argument, default "all"):
#   tests     the whole device suite
#   bench     bench lines of every workload (north star with the CPU leg) + the config-3 end-to-end figure
#   prof      rocprofv3 kernel statistics of the north-star / cfg3 / recurrent workloads, HBM counters (separate --pmc passes
#             with --kernel-trace only, as MI355X_MICROARCH.md prescribes), SQ counters
#   proxy     single-GPU shard proxies of the data-parallel jobs (north star, cfg4 = SMAC, cfg5 = Hanabi)
# Everything lands under gpurun_out/r06/ (python tools/summarize_round.py r06 copies the summaries into profiles/).
set -u
STAGE=${1:-all}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p $OUT
# provenance of everything below (the box has no .git): the commit comes with the command line (MAPPO_COMMIT=... bash tools/profile_r06.sh),
# the digest is taken from the kernel sources as they are here
echo "${MAPPO_COMMIT:-unknown}" > $OUT/commit.txt
python -c "import bench; print(bench.csrc_digest())" > $OUT/csrc_digest.txt 2>/dev/null
export TMPDIR=/tmp
ulimit -c 0
cd $REPO

if [[ $STAGE == all || $STAGE == tests ]]; then
  rm -f gpurun_out/parity_margins.json
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
  cp gpurun_out/parity_margins.json $OUT/parity_margins.json 2>/dev/null
fi

if [[ $STAGE == all || $STAGE == bench ]]; then
  timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench_ns.json      # (the driver's command line: carries `workloads`)
  # the same workloads as their own command lines (what `workloads` must agree with within +- 3 %)
  timeout 300 python bench.py --workload cfg2 --steps 30 --warmup 5 --no-f32-mfma 2>&1 | tail -1 > $OUT/bench_cfg2.json
  timeout 300 python bench.py --workload cfg3 --steps 10 --warmup 2 --no-f32-mfma 2>&1 | tail -1 > $OUT/bench_cfg3.json
  timeout 300 python bench.py --workload ns_rnn --steps 3 --warmup 1 --no-f32-mfma --cpu-sample-threads 16 2>&1 | tail -1 > $OUT/bench_ns_rnn.json
  timeout 300 python bench.py --workload smac --steps 10 --warmup 2 --no-f32-mfma 2>&1 | tail -1 > $OUT/bench_smac.json
  timeout 300 python bench.py --workload smac --threads 64 --steps 40 --warmup 5 --no-f32-mfma --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_smac_shard64.json
  timeout 500 python bench.py --workload hanabi --steps 3 --warmup 1 --no-f32-mfma --cpu-sample-threads 8 2>&1 | tail -1 > $OUT/bench_hanabi.json
  timeout 600 python tools/cfg3_end_to_end.py --out $OUT/cfg3_end_to_end.json > $OUT/cfg3_end_to_end.log 2>&1
  tail -1 $OUT/cfg3_end_to_end.log | cut -c1-400
fi

if [[ $STAGE == all || $STAGE == prof ]]; then
  cd /tmp
  # (MAPPO_TWO_STREAM_UPDATE=0: cfg2 / smac / hanabi evaluate actor and critic on two streams, and a kernel that shares the chip
  # has no duration of its own -- the statistics are of the launches one after the other, like bench.py's roofline timings;
  # ns / cfg3 / ns_rnn run on one stream anyway)
  for w in ns cfg2 cfg3 ns_rnn smac hanabi; do
    MAPPO_TWO_STREAM_UPDATE=0 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_$w -o $w -- python $REPO/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-f32-mfma > $OUT/prof_$w.log 2>&1
  done
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pmc_$C -o ns -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32-mfma > $OUT/pmc_$C.log 2>&1
  done
  cd $REPO
  MAPPO_ROUND=r06 bash tools/pmc_sq_pass.sh ns > $OUT/pmc_sq_ns.txt 2>&1
  MAPPO_ROUND=r06 bash tools/pmc_sq_pass.sh ns_rnn > $OUT/pmc_sq_ns_rnn.txt 2>&1
  cd $REPO
fi

if [[ $STAGE == all || $STAGE == proxy ]]; then
  timeout 1500 python tools/shard_proxy.py --out $OUT/shard_proxy.json > $OUT/shard_proxy.log 2>&1; tail -1 $OUT/shard_proxy.log | cut -c1-600
fi

find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +30M -delete
ls $OUT; cut -c1-300 $OUT/bench_ns.json 2>/dev/null
