#!/bin/bash
# tools/ab_sorted.sh [variant ...] : C3 / C4 / C5 at the spp of record, three lanes + exclusive pass ("tree" = the in-tree library)
for v in "$@"; do
  for c in c3:1024 c4:156 c5:162; do
    cfg=${c%%:*}; spp=${c##*:}
    if [ "$v" = "tree" ]; then unset ADAPT_MI_LIB; else export ADAPT_MI_LIB=$PWD/build_exp/libadapt_mi_$v.so; fi
    python bench.py --config $cfg --steps 1 --warmup 1 --spp $spp --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err
    python - <<PY
import json
try:
    d = json.load(open("/tmp/ab.json")); pk = d["roofline"]["per_kernel"]
    print("$v $cfg", d["value"], d["roofline"].get("one_lane_Msamples/s"), {k: v["ms"] for k, v in pk.items() if k in ("extend", "shadow", "shade")})
except Exception as e:
    print("$v $cfg failed", e, open("/tmp/ab.err").read()[-300:])
PY
  done
done
