#!/bin/bash
# First gpurun call of the next round: the whole device suite, the bench lines of every workload + config 3 end to end, the
# kernel statistics / PMC / SQ passes and the shard proxies on the current code (tools/profile_r04.sh; results under
# gpurun_out/r04/, summarise with `python tools/summarize_round.py r04`).
#   gpurun --timeout 3000 -- 'bash tools/next_round_first_call.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
bash tools/profile_r04.sh all
