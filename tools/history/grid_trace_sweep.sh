run() { env $ENVS python bench.py --config $CFG --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline --no-exclusive-pass --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG $ENVS', d['value'])"; }
for cs in c2:768 c1:192 c3:256 v1:128; do CFG=${cs%%:*}; SPP=${cs##*:}
for t in 0 3 4 5 6; do if [ $t = 0 ]; then ENVS="A=1" run; else ENVS="APT_GRID_TRACE=$t APT_GRID_SHADOW=$t" run; fi; done; done
