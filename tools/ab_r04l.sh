#!/bin/bash
# GPU box: (a) K12 forward with packed gate / LayerNorm math against the library of the previous commit (recurrent workloads);
# (b) the 12-wave version-3 forward for narrow inputs (MAPPO_MLP_FLAGS=24) at the north star; (c) the library GEMM at the
# dW1 shape (tools/probe_dw1_gemm.py).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_l
mkdir -p $OUT
cd $REPO
OLD=$REPO/on-policy_amd/lib/libmappo_hip_OLD.so
timeout 600 python -m pytest tests/test_gpu_gru_seq.py tests/test_gpu_trainer_h64.py -q > $OUT/tests_new.log 2>&1; tail -2 $OUT/tests_new.log
for w in ns_rnn smac ns_rnn; do
  MAPPO_HIP_LIB=$OLD timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_old.jsonl
  timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_new.jsonl
done
for i in 1 2; do
  timeout 300 python bench.py --workload ns --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_ns_f0.jsonl
  MAPPO_MLP_FLAGS=24 timeout 300 python bench.py --workload ns --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_ns_f24.jsonl
done
timeout 200 python tools/probe_dw1_gemm.py > $OUT/dw1_gemm.json 2>&1; tail -1 $OUT/dw1_gemm.json
timeout 200 python tools/probe_dw1_gemm.py --rows 2621440 >> $OUT/dw1_gemm.json 2>&1; tail -1 $OUT/dw1_gemm.json
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_l/"
for name in ("bench_old", "bench_new", "bench_ns_f0", "bench_ns_f24"):
    for l in open(out + name + ".jsonl"):
        if l.startswith("{"):
            d = json.loads(l); print(name, d["config"]["workload"][:40], d["ms_per_step"])
PY
