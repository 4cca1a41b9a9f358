#!/bin/bash
# GPU box: K12 with the backward's r / z blocks in six-term form too (option bit 1024) -- device tests under both forms,
# the reference's recurrent fixtures under all opt-in bits, recurrent north star and SMAC shapes default against 1856.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_y
mkdir -p $OUT
cd $REPO
timeout 120 python -m pytest tests/test_gpu_gru_seq.py -q -p no:cacheprovider > $OUT/gpu_gru_both.log 2>&1
echo "test_gpu_gru_seq (both forms) rc=$?"; tail -2 $OUT/gpu_gru_both.log
MAPPO_MLP_FLAGS=1856 timeout 120 python -m pytest tests/test_gpu_trainer_h64.py tests/test_gpu_device_sampler_route.py tests/test_gpu_runners.py -q -p no:cacheprovider > $OUT/gpu_fixtures_flag1856.log 2>&1
echo "fixtures under flags 1856 rc=$?"; tail -2 $OUT/gpu_fixtures_flag1856.log
for w in ns_rnn smac; do
  for f in 0 1856; do
    MAPPO_MLP_FLAGS=$f timeout 100 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-six-term 2>&1 | tail -1 >> $OUT/${w}_flag$f.jsonl
  done
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_y/"
for w in ("ns_rnn", "smac"):
    for f in (0, 1856):
        rows = [json.loads(l) for l in open(out + "%s_flag%d.jsonl" % (w, f)) if l.startswith("{")]
        print(w, f, [r["ms_per_step"] for r in rows], [r["value"] for r in rows])
PY
