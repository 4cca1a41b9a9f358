#!/bin/bash
# ONE parametrised A / B harness for the GPU box (replaces the 25 one-off tools/ab_r04*.sh of round 4, which live on in the git
# history; what they measured is recorded in profiles/r04_ab_*.json).  Alternates the variants inside ONE gpurun call, because
# the boxes of the pool differ by +-2 %, and prints ms per step per variant.
#
#   tools/ab.sh NAME "WORKLOAD [bench.py flags]" REPS VARIANT [VARIANT ...]
#
#   NAME      output directory gpurun_out/ab/NAME (one <variant>.jsonl per variant: bench.py's JSON lines)
#   WORKLOAD  what follows `python bench.py --workload`, e.g. "smac --threads 64"
#   VARIANT   label[:ENV=VALUE[,ENV=VALUE...]]; the environment that distinguishes the variant, e.g.
#               graph:MAPPO_UPDATE_GRAPH=1  eager:MAPPO_UPDATE_GRAPH=0
#               six_term:MAPPO_MATRIX_ARITHMETIC=six_term  f32:MAPPO_MATRIX_ARITHMETIC=f32_mfma
#               new  old:MAPPO_HIP_LIB=/path/to/libmappo_hip_OLD.so        (tools/ab_old_lib.sh builds the old library)
#               k15:MAPPO_LINEAR512=1  library:MAPPO_LINEAR512=0
#   STEPS / WARMUP (environment): bench.py --steps / --warmup, default 6 / 2.
#
#   gpurun --timeout 900 -- 'tools/ab.sh graph_smac64 "smac --threads 64" 2 graph:MAPPO_UPDATE_GRAPH=1 eager:MAPPO_UPDATE_GRAPH=0'
set -u
[ $# -ge 4 ] || { sed -n 2,20p "$0"; exit 2; }
NAME=$1; WORK=$2; REPS=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ab/$NAME
mkdir -p "$OUT"
cd "$REPO"
for rep in $(seq 1 "$REPS"); do
  for v in "$@"; do
    label=${v%%:*}
    envs=""
    [ "$v" != "$label" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
    env $envs timeout 600 python bench.py --workload $WORK --steps ${STEPS:-6} --warmup ${WARMUP:-2} --no-cpu-baseline --no-f32-mfma 2>&1 | tail -1 >> "$OUT/$label.jsonl"
  done
done
python - "$OUT" <<'PY'
import glob, json, os, sys
for p in sorted(glob.glob(os.path.join(sys.argv[1], "*.jsonl"))):
    rows = [json.loads(l) for l in open(p) if l.startswith("{")]
    print("%-24s ms/step %s  env-steps/s %s" % (os.path.basename(p)[:-6], [r["ms_per_step"] for r in rows], [round(r["value"]) for r in rows]))
PY
