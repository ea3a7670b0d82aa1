run() { env $ENVS python bench.py --config c2 --steps 2 --warmup 1 --spp 768 --no-cpu-baseline --no-exclusive-pass --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$ENVS', d['value'])"; }
ENVS="A=1" run
for s in 2 3 4; do for t in 2 3 4 6; do ENVS="APT_GRID_SMALL=$s APT_GRID_TRACE=$t APT_GRID_SHADOW=$t" run; done; done
for l in 4 5 6; do ENVS="APT_GRID_SMALL=2 APT_GRID_TRACE=3 APT_GRID_SHADOW=3 APT_LANES=4" run; break; done
ENVS="APT_GRID_SMALL=2 APT_GRID_TRACE=2 APT_GRID_SHADOW=2 APT_LANES=4" run
ENVS="APT_GRID_SMALL=1 APT_GRID_TRACE=2 APT_GRID_SHADOW=2 APT_LANES=4" run
