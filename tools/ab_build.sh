#!/bin/bash
# Tuning aid: builds a variant of libmappo_hip.so with extra -D flags for mappo_mlp.hip into on-policy_amd/lib/libmappo_hip_<tag>.so
# (select it with MAPPO_HIP_LIB=...), so that two variants can be timed in the same gpurun call on the same box.
#   tools/ab_build.sh B -DMAPPO_MLP_NO_CONTRACT
set -e
TAG=$1; shift
cd "$(dirname "$0")/../on-policy_amd/csrc"
make -s >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include "$@" -c mappo_mlp.hip -o /tmp/mappo_mlp_$TAG.o
OBJS=$(ls *.o | grep -v mappo_mlp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/mappo_mlp_$TAG.o -o ../lib/libmappo_hip_$TAG.so
echo ../lib/libmappo_hip_$TAG.so
