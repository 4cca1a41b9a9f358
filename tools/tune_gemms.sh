#!/bin/bash
# Runs on the MI355X box: tunes the GEMM shapes of the shipped workloads that still go through the BLAS libraries (the
# hidden-64 MLP trunks do not since K9: recurrent north star incl. its 8-GPU shard, SMAC incl. 64 threads per GPU, cfg2,
# Hanabi's hidden-512 networks) and leaves TunableOp's result file under gpurun_out/gemm_tuning/; merge it into
# on-policy_amd/onpolicy/tuned_gemms_gfx950.csv with tools/merge_tuned_gemms.py afterwards.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export MAPPO_GEMM_TUNING_CACHE=$REPO/gpurun_out/gemm_tuning
rm -rf $MAPPO_GEMM_TUNING_CACHE
cd $REPO
run() { timeout 260 python bench.py "$@" --no-cpu-baseline --steps 1 --warmup 0 2>&1 | tail -1 | cut -c1-60; }
run --workload ns_rnn
run --workload ns_rnn --threads 512
run --workload smac
run --workload smac --threads 64
run --workload cfg2
# (hanabi: its hidden-512 shapes are in the shipped table since round 2; run --workload hanabi to refresh them)
wc -l $MAPPO_GEMM_TUNING_CACHE/*
