#!/usr/bin/env python3
"""Rows of BASELINE.md §4 from the committed bench lines (profiles/r0N_bench_<config>.json): python tools/baseline_table.py [round]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ROUND1 = {"c1": "2 851", "c2": "2 341", "c3": "615", "c4": "816", "c5": "731", "v1": "804-816", "v2": "528-531", "v3": "543"}
NAMES = {
    "c1": "C1 cbox 256x256, 64 spp, 4 b", "c2": "C2 cbox 512x512, 1024 spp, 8 b", "c3": "C3 csphere 512x512, 1024 spp, 16 b, S=4",
    "c4": "C4 three-bunnies* 800x800, 8 b, S=2 (128 spp/step)", "c5": "C5 bunny-field* 1280x720, 16 b, S=1 (64 spp/step)",
    "v1": "V1 fog Cornell box (the reference's `scenes/vpt/cbox.xml` set-up), 512x512, 256 spp, 16 b, **volumetric tracer**",
    "v2": "V2 media_a (all surface models, three media), 512x512, 256 spp, 8 b, S=2, **volumetric tracer**, sorted by event class",
    "v3": "V3 volgrid_a (20x16x12 RGB grid volume), 512x512, 128 spp, 8 b, S=2, **volumetric tracer**, sorted by event class",
}
print("| Config | CPU 128T (8T) Msamples/s | round 1, 1xMI355X | **round 2, 1xMI355X** (3 render lanes; V1 / V2: 4) | one lane | k_extend alone | k_shade alone | k_shadow alone | whole pipeline, algorithmic GB/s | relMSE vs CPU | pixels within 1e-3(1+x) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for c in ("c1", "c2", "c3", "c4", "c5", "v1", "v2", "v3"):
    d = json.load(open(os.path.join(ROOT, "profiles", f"r{rnd:02d}_bench_{c}.json")))
    r, cb, par = d["roofline"], d["cpu_baseline"], d["parity"]
    def stage(k):
        s = r["stages"][k]
        v = s.get("valu", {}).get("busy_frac")
        return f"{s['GB/s']:.0f} ({100 * s['frac']:.1f} %" + (f"; VALU {100 * v:.0f} %" if v is not None else "") + ")"
    print(f"| {NAMES[c]} | {cb['value']:.1f} ({cb.get('at_8_threads', {}).get('value', float('nan')):.1f}) | {ROUND1[c]} | **{d['value']:.0f}** | {r['one_lane_Msamples/s']:.1f} | "
          f"{stage('extend')} | {stage('shade')} | {stage('shadow')} | {r['pipeline_GB/s']:.0f} ({100 * r['pipeline_frac']:.1f} %) | {par['relMSE']:.1e} | {100 * par['frac_within_1e-3']:.3f} % |")
