#!/bin/bash
# tools/fastmath_ab.sh "<variants>" "<configs>"   price the arithmetic variants under build_exp/ ("-" = the in-tree library):
# three-lane rate, one-lane per-kernel ms and the parity leg against the oracle on the same Philox stream.
VARS=${1:-"- fma em0 div all"}; CFGS=${2:-"c2 c3"}
mkdir -p gpurun_out/fastmath
for v in $VARS; do
  for cfg in $CFGS; do
    if [ "$v" = "-" ]; then unset ADAPT_MI_LIB; name=exact; else export ADAPT_MI_LIB=$PWD/build_exp/libadapt_mi_$v.so; name=$v; fi
    python bench.py --config $cfg --steps 3 --warmup 1 --cpu-seconds 8 > gpurun_out/fastmath/${name}_$cfg.json 2> gpurun_out/fastmath/${name}_$cfg.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/fastmath/${name}_$cfg.json")); r = d["roofline"]; p = d.get("parity", {})
    print("$name $cfg", d["value"], "one-lane", r.get("one_lane_Msamples/s"), {k: v["ms"] for k, v in r["per_kernel"].items() if k in ("extend", "shadow", "shade")},
          "relMSE %.3g within %.6f max %.3g" % (p.get("relMSE", -1), p.get("frac_within_1e-3", -1), p.get("max_abs", -1)), d["per_sample"])
except Exception as e:
    print("$name $cfg failed", e, open("gpurun_out/fastmath/${name}_$cfg.err").read()[-400:])
PY
  done
done
