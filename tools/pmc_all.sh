#!/bin/bash
# every bench config's PMC / kernel-trace passes for this round (GPU box): tools/pmc_all.sh [round]
RND=${1:-2}
for c in c2:64 c1:64 c3:32 c4:32 c5:16 v1:32 v2:32 v3:32; do
  bash tools/pmc_round.sh ${c%%:*} ${c##*:} $RND 2>&1 | tail -1
done
