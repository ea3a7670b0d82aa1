#!/bin/bash
# tools/grid_sweep.sh   persistent-grid size of the streaming stages (workgroups per CU) per config; prints Msamples/s without per-launch events
run() { cfg=$1; spp=$2; shift 2
  env $ENVS python bench.py --config $cfg --steps 2 --warmup 1 --spp $spp --no-cpu-baseline --no-exclusive-pass --no-profile "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$ENVS', '$*', d['value'])"
}
for cs in c2:768 c3:384 c1:192 c4:96 c5:48 v1:96; do
  for g in 2 3 4 5 8; do ENVS="APT_GRID_SMALL=$g" run ${cs%%:*} ${cs##*:}; done
done
