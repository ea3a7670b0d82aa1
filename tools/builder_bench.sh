#!/bin/bash
# traversal speed on the trees of the three builders (GPU box): tools/builder_bench.sh "<configs>"
for c in ${1:-"c4 c5"}; do for b in sah lbvh ploc; do
  v=$(APT_BVH_BUILDER=$b python bench.py --config $c --steps 1 --warmup 1 --spp 64 --no-cpu-baseline --no-exclusive-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity',{}).get('frac_within_1e-3'))")
  echo "$c builder $b: $v"
done; done
