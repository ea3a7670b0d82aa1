#!/bin/bash
# A/B of an environment knob on chosen bench configs: tools/gpu_env_probe.sh "c4 c5" APT_DYN_FETCH 0 1
CFGS=$1; VAR=$2; shift 2
for val in "$@"; do
  for cfg in $CFGS; do
    extra=""; [ "$cfg" = "c4" ] && extra="--spp 128"; [ "$cfg" = "c5" ] && extra="--spp 64"
    env $VAR=$val python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$VAR=$val', '$cfg', d['value'], r.get('one_lane_Msamples/s'), {k:(v['ms'],v['launches']) for k,v in r['per_kernel'].items() if k in ('extend','shade','shadow')})"
  done
done
