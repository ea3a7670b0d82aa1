#!/bin/bash
# tools/ab_run.sh <cfg:spp> "<ENV=.. ...>" ...   one line per environment set: three-lane rate, one-lane rate, one-lane per-kernel ms.
# ("-" = no extra environment; ADAPT_MI_LIB=build_exp/libadapt_mi_<x>.so selects a variant build)
cs=$1; shift
cfg=${cs%%:*}; spp=${cs##*:}
for envs in "$@"; do
  [ "$envs" = "-" ] && envs=""
  env $envs python bench.py --config $cfg --steps 2 --warmup 1 --spp $spp --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err
  python - <<PY
import json
try:
    d = json.load(open("/tmp/ab.json")); pk = d["roofline"]["per_kernel"]
    print("$cfg [$envs]", d["value"], "one-lane", d["roofline"].get("one_lane_Msamples/s"), {k: v["ms"] for k, v in pk.items() if k in ("extend", "shadow", "shade")}, d["config"]["traversal"])
except Exception as e:
    print("$cfg [$envs] failed", e, open("/tmp/ab.err").read()[-300:])
PY
done
