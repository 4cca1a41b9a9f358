#!/bin/bash
# Tuning aid for the tools/ab_r04*.sh scripts: builds libmappo_hip.so of another commit (default: HEAD) in a temporary worktree and
# leaves it as on-policy_amd/lib/libmappo_hip_OLD.so (git-ignored; select it with MAPPO_HIP_LIB=...), so that the library of the
# working tree and that of the commit can be timed alternating inside one gpurun call on one box.
#   tools/ab_old_lib.sh [commit]
set -e
REV=${1:-HEAD}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
WT=$(mktemp -d /tmp/mappo_old_XXXX)
git -C "$ROOT" worktree add -q --detach "$WT" "$REV"
make -s -C "$WT/on-policy_amd/csrc" >/dev/null
cp "$WT/on-policy_amd/lib/libmappo_hip.so" "$ROOT/on-policy_amd/lib/libmappo_hip_OLD.so"
git -C "$ROOT" worktree remove --force "$WT"
echo "$ROOT/on-policy_amd/lib/libmappo_hip_OLD.so  <-  $(git -C "$ROOT" rev-parse --short "$REV")"
