#!/bin/bash
# A/B of environment knobs on one config (GPU box): tools/ab_env.sh <config> "<VAR=val ...>" ["<VAR=val ...>" ...]   (first set may be "")
CFG=$1; shift
for envs in "$@"; do
  for rep in 1 2; do
    v=$(env $envs python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config'].get('shade_variant'))")
    echo "$CFG [$envs]: $v"
  done
done
