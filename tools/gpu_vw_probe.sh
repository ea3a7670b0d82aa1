#!/bin/bash
# volumetric bench under library variants (ADAPT_MI_LIB): one line per variant
for lib in adapt_amd/libadapt_mi.so tools/_lib_vw1.so tools/_lib_vw2.so tools/_lib_vw4.so; do
  for cfg in v1; do
    ADAPT_MI_LIB=$PWD/$lib python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', '$cfg', d['value'], r.get('one_lane_Msamples/s'), d['config']['shade_variant'], {k:(v['ms'],v['GB/s']) for k,v in r['per_kernel'].items()})"
  done
done
