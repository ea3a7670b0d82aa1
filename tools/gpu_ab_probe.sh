#!/bin/bash
# A/B of two builds of the library on chosen bench configs: tools/gpu_ab_probe.sh "c4 c5" libA.so libB.so
CFGS=${1:-"c4 c5"}; shift
for lib in "$@"; do
  for cfg in $CFGS; do
    extra=""; [ "$cfg" = "c4" ] && extra="--spp 128"; [ "$cfg" = "c5" ] && extra="--spp 64"
    ADAPT_MI_LIB=$PWD/$lib python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', '$cfg', d['value'], r.get('one_lane_Msamples/s'), {k:(v['ms'],v['launches']) for k,v in r['per_kernel'].items() if k in ('extend','shade','shadow')})"
  done
done
