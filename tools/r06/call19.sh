#!/bin/bash
# round 6, GPU call 19: slab standardisation inside K2's launch, the full pass through the same code
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_standardize_at_insert.py tests/test_gpu_parity.py tests/test_gpu_runners.py tests/test_gpu_rollout_graph.py tests/test_gpu_mpe_end_to_end.py tests/test_gpu_separated.py tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py tests/test_gpu_cfg_shapes.py tests/test_gpu_mid_size.py tests/test_gpu_update_graph.py -m gpu -q 2>&1 | tail -8 > gpurun_out/call19_tests.log
timeout 600 python tools/cfg3_end_to_end.py --out gpurun_out/call19_cfg3_e2e.json > gpurun_out/call19_cfg3.log 2>&1
MAPPO_STANDARDIZE_AT_INSERT=0 timeout 600 python tools/cfg3_end_to_end.py --out gpurun_out/call19_cfg3_e2e_std_at_train.json > gpurun_out/call19_cfg3b.log 2>&1
timeout 600 python tools/cfg3_end_to_end.py --out gpurun_out/call19_cfg3_e2e_b.json > gpurun_out/call19_cfg3c.log 2>&1
python - <<'PY'
import json
for n in ("call19_cfg3_e2e","call19_cfg3_e2e_std_at_train","call19_cfg3_e2e_b"):
    c=json.load(open('gpurun_out/%s.json'%n)); print(n, c["env_steps_per_s_rollout_plus_update"], c["rollout_ms_per_env_step"], c["update_s"])
PY
timeout 300 python bench.py --no-cpu-baseline --no-workloads --no-f32-mfma --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
tail -4 gpurun_out/call19_tests.log
