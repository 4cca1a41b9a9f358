#!/bin/bash
# round 6, GPU call 10: K15 weight gradient with the X tile's bf16 planes shared through LDS (tuning bit 8 = round 5's form, every
# wave splits all four column tiles itself) -- parity, then A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_lin512.py tests/test_gpu_cfg_shapes.py -m gpu -q 2>&1 | tail -8 > gpurun_out/call10_tests.log
MAPPO_MLP_FLAGS=8 timeout 600 python -m pytest tests/test_gpu_lin512.py -m gpu -q 2>&1 | tail -4 >> gpurun_out/call10_tests.log
timeout 300 python tools/bench_lin512.py > gpurun_out/call10_lin512_shared.json 2>/dev/null
MAPPO_MLP_FLAGS=8 timeout 300 python tools/bench_lin512.py > gpurun_out/call10_lin512_own.json 2>/dev/null
: > gpurun_out/call10_k15.txt
for f in 0 8 0 8; do
  MAPPO_MLP_FLAGS=$f timeout 600 python bench.py --workload hanabi --no-cpu-baseline --no-f32-mfma --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('flags $f step', d['ms_per_step'], 'value', d['value'], 'K15 fwd', r['launch_ms'], r['frac'], 'wgrad', d['roofline_linear512_wgrad']['launch_ms'], d['roofline_linear512_wgrad']['frac'])" >> gpurun_out/call10_k15.txt
done
cat gpurun_out/call10_k15.txt; cat gpurun_out/call10_tests.log | tail -8
python - <<'PY'
import json
for n in ("shared","own"):
    d=json.loads(open('gpurun_out/call10_lin512_%s.json'%n).read().strip().splitlines()[-1])
    print(n, [(r["K"], r["k15_wgrad"]) for r in d["runs"]])
PY
