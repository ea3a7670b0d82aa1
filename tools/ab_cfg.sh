#!/bin/bash
# tools/ab_cfg.sh <config:spp> <lib|-> ...   one-lane per-kernel ms + 3-lane value for each library ("-" = the in-tree one)
cs=$1; shift
cfg=${cs%%:*}; spp=${cs##*:}
for v in "$@"; do
  if [ "$v" = "-" ]; then unset ADAPT_MI_LIB; else export ADAPT_MI_LIB=$PWD/build_exp/libadapt_mi_$v.so; fi
  python bench.py --config $cfg --steps 1 --warmup 1 --spp $spp --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err
  python - <<PY
import json
try:
    d = json.load(open("/tmp/ab.json")); pk = d["roofline"]["per_kernel"]
    print("$v $cfg", d["value"], "one-lane", d["roofline"].get("one_lane_Msamples/s"), {k: v["ms"] for k, v in pk.items() if k in ("extend", "shadow", "shade")})
except Exception as e:
    print("$v $cfg failed", e, open("/tmp/ab.err").read()[-300:])
PY
done
