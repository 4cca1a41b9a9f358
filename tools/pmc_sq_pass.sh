#!/bin/bash
# Runs on the MI355X box: one SQ / GRBM counter pass over the north-star bench step (separate --pmc pass with
# --kernel-trace only, as MI355X_MICROARCH.md prescribes) -> per kernel: MFMA busy cycles, wave / wait / active cycles and
# GRBM_GUI_ACTIVE (shader-clock cycles the GPU was busy), i.e. the clock the chip ran at and the matrix pipe's busy share.
# python tools/pmc_sq_summary.py summarises into profiles/r03_pmc_sq_summary.json.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-ns}                      # bench workload (ns_rnn: the K12 kernels)
OUT=$REPO/gpurun_out/${MAPPO_ROUND:-r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq_$W -o $W -- python $REPO/bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq_$W.log 2>&1
find $OUT/pmc_sq_$W -name "*.db" -delete
ls -la $OUT/pmc_sq_$W; tail -3 $OUT/pmc_sq_$W.log | cut -c1-200
