#!/bin/bash
# GPU box: the six-term weight-gradient kernel with the previous tile's MFMAs between a tile's LDS reads and their split
# (new library) against the unpipelined form (libmappo_hip_OLD.so), option bits 64 + 256 + 512, alternating on one box.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04/ab_u
mkdir -p $OUT
cd $REPO
OLD=$REPO/on-policy_amd/lib/libmappo_hip_OLD.so
timeout 300 python -m pytest tests/test_gpu_mlp.py -q -p no:cacheprovider > $OUT/gpu_mlp_both.log 2>&1
echo "test_gpu_mlp rc=$?"; tail -2 $OUT/gpu_mlp_both.log
for i in 1 2; do
  MAPPO_HIP_LIB=$OLD MAPPO_MLP_FLAGS=832 timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-six-term 2>&1 | tail -1 >> $OUT/ns_old.jsonl
  MAPPO_MLP_FLAGS=832 timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-six-term 2>&1 | tail -1 >> $OUT/ns_new.jsonl
done
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04/ab_u/"
for name in ("ns_old", "ns_new"):
    rows = [json.loads(l) for l in open(out + name + ".jsonl") if l.startswith("{")]
    print(name, [r["ms_per_step"] for r in rows], [r["roofline_mlp_backward"]["launch_ms"] for r in rows])
PY
