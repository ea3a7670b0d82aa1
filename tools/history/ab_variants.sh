#!/bin/bash
# A/B the libraries under build_exp/ on C4 / C5 (one lane, per-kernel ms): tools/ab_variants.sh name1 name2 ...
for v in "$@"; do
  for c in c4:64 c5:36; do
    cfg=${c%%:*}; spp=${c##*:}
    ADAPT_MI_LIB=$PWD/build_exp/libadapt_mi_$v.so python bench.py --config $cfg --steps 1 --warmup 1 --spp $spp --no-cpu-baseline --no-exclusive-pass --lanes 1 > /tmp/ab.json 2>/tmp/ab.err
    python - <<PY
import json
try:
    d = json.load(open("/tmp/ab.json")); pk = d["roofline"]["per_kernel"]
    print("$v $cfg", d["value"], {k: v["ms"] for k, v in pk.items() if k in ("extend", "shadow", "shade")})
except Exception as e:
    print("$v $cfg failed", e, open("/tmp/ab.err").read()[-300:])
PY
  done
done
