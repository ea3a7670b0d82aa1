#!/usr/bin/env python3
"""tools/bl.py <bench json>...  -  one line per bench result: value, one-lane rate, per-kernel ms of the exclusive pass"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f, d["value"], "one-lane", r.get("one_lane_Msamples/s"), {k: round(v["ms"], 2) for k, v in r["per_kernel"].items()}, "parity", {k: d.get("parity", {}).get(k) for k in ("relMSE", "frac_within_1e-3", "non_finite_pixels_coincide")})
    except Exception as e:
        print(f, "failed:", e)
