run() { # cfg spp extra-args... ; env via ENVS
  cfg=$1; spp=$2; shift 2
  env $ENVS python bench.py --config $cfg --steps 2 --warmup 1 --spp $spp --no-cpu-baseline --no-exclusive-pass --no-profile "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '$ENVS', '$*', d['value'], d['config'].get('spp_per_batch'))"
}
for l in 1 2 3 4; do ENVS="A=1" run c2 768 --lanes $l; done
for b in 32 64 128 256; do ENVS="A=1" run c2 768 --spp-per-batch $b; done
for l in 2 3 4; do ENVS="A=1" run c3 384 --lanes $l; done
for b in 16 32 64 128; do ENVS="A=1" run c3 384 --spp-per-batch $b; done
for g in 4 6 8 12; do ENVS="APT_GRID_TRACE=$g" run c2 768; done
for g in 4 6 8 12; do ENVS="APT_GRID_SMALL=$g" run c2 768; done
