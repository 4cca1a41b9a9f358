#!/bin/bash
# round 6, GPU call 18: the observation slabs' standardised copies recomputed inside K2's own launch (mappo_slab_copy_std)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_standardize_at_insert.py tests/test_gpu_parity.py tests/test_gpu_runners.py tests/test_gpu_rollout_graph.py tests/test_gpu_mpe_end_to_end.py tests/test_gpu_separated.py -m gpu -q 2>&1 | tail -8 > gpurun_out/call18_tests.log
timeout 600 python tools/cfg3_end_to_end.py --out gpurun_out/call18_cfg3_e2e.json > gpurun_out/call18_cfg3.log 2>&1
MAPPO_STANDARDIZE_AT_INSERT=0 timeout 600 python tools/cfg3_end_to_end.py --out gpurun_out/call18_cfg3_e2e_std_at_train.json > gpurun_out/call18_cfg3b.log 2>&1
python - <<'PY'
import json
for n in ("call18_cfg3_e2e","call18_cfg3_e2e_std_at_train"):
    c=json.load(open('gpurun_out/%s.json'%n)); print(n, c["env_steps_per_s_rollout_plus_update"], c["rollout_ms_per_env_step"], c["update_s"])
PY
tail -4 gpurun_out/call18_tests.log
