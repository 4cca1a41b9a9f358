#!/bin/bash
# Runs on the MI355X box: tunes the GEMM shapes of the shipped workloads (per-rank sizes of the 1 / 2 / 4 / 8 GPU
# north-star runs, the recurrent north star, SMAC and cfg2 shapes) and merges the winners into
# on-policy_amd/tuned_gemms_gfx950.csv (copy it from gpurun_out/ afterwards).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export MAPPO_GEMM_TUNING_CACHE=$REPO/gpurun_out/gemm_tuning
rm -rf $MAPPO_GEMM_TUNING_CACHE
cd $REPO
for t in 4096 2048 1024 512; do
  python bench.py --threads $t --no-cpu-baseline --steps 1 --warmup 0 2>&1 | tail -1 | cut -c1-60
done
python bench.py --workload ns_rnn --no-cpu-baseline --steps 1 --warmup 0 2>&1 | tail -1 | cut -c1-60
python bench.py --workload smac --no-cpu-baseline --steps 1 --warmup 0 2>&1 | tail -1 | cut -c1-60
python bench.py --workload smac --threads 64 --no-cpu-baseline --steps 1 --warmup 0 2>&1 | tail -1 | cut -c1-60
python bench.py --workload cfg2 --no-cpu-baseline --steps 1 --warmup 0 2>&1 | tail -1 | cut -c1-60
wc -l $MAPPO_GEMM_TUNING_CACHE/*
