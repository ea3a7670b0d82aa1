#!/bin/bash
# volumetric bench configs, one summary line each (optionally under ADAPT_MI_LIB=<other build>)
for cfg in v1 v2; do
  python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$cfg', d['value'], r.get('one_lane_Msamples/s'), d['per_sample'], {k:(v['ms'],v['launches'],v['GB/s']) for k,v in r['per_kernel'].items() if k in ('extend','shade','shadow')})"
done
