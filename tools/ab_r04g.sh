#!/bin/bash
# GPU box: mappo_mlp.hip compiled with -mllvm -amdgpu-mfma-vgpr-form (MFMA accumulators in the VGPR half: fewer AGPR <-> VGPR
# moves in the backward chain, 816 -> 591 static) against the default build, alternating on one box.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_g
mkdir -p $OUT
cd $REPO
VF=$REPO/on-policy_amd/lib/libmappo_hip_VF.so
MAPPO_HIP_LIB=$VF timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py tests/test_gpu_gru_seq.py -q > $OUT/tests_vf.log 2>&1; tail -2 $OUT/tests_vf.log
for i in 1 2 3; do
  timeout 200 python tools/bench_mlp.py --sequential --reps 7 >> $OUT/mlp_def.jsonl 2>&1
  MAPPO_HIP_LIB=$VF timeout 200 python tools/bench_mlp.py --sequential --reps 7 >> $OUT/mlp_vf.jsonl 2>&1
done
for w in ns cfg3 ns_rnn; do
  timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_def.jsonl
  MAPPO_HIP_LIB=$VF timeout 300 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 >> $OUT/bench_vf.jsonl
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_g/"
for name in ("mlp_def", "mlp_vf"):
    rows = [json.loads(l) for l in open(out + name + ".jsonl") if l.startswith("{")]
    for din in (384, 48):
        print(name, "din", din, "fwd_ms", [r["fwd_ms"] for r in rows if r["din"] == din], "bwd_ms", [r["bwd_ms"] for r in rows if r["din"] == din])
for name in ("bench_def", "bench_vf"):
    for l in open(out + name + ".jsonl"):
        if l.startswith("{"):
            d = json.loads(l); print(name, d["config"]["workload"][:36], d["ms_per_step"], "bwd", d["roofline_mlp_backward"]["launch_ms"])
PY
