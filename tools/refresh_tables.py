"""Rewrite the per-config rows of BASELINE.md §4 and the stage rows of DESIGN.md §6 from profiles/r0N_bench_*.json and r0N_*_counters.json.
    python tools/refresh_tables.py [round]"""
import json, re, sys
rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 6
P = f"profiles/r{rnd:02d}"
rows = [("c1", "C1 cbox 256x256, 64 spp, 4 b", "2 851", "2 770", "3 528", "4 883", "4 967"), ("c2", "C2 cbox 512x512, 1024 spp, 8 b", "2 341", "2 372", "3 597", "4 161", "4 502"), ("c3", "C3 csphere 512x512, 1024 spp, 16 b, S=4", "615", "614", "973", "1 086", "1 208"),
        ("c4", "C4 three-bunnies* 800x800, 8 b, S=2 (156 spp/step)", "816", "1 329", "1 346", "1 824", "1 939"), ("c5", "C5 bunny-field* 1280x720, 16 b, S=1 (216 spp/step)", "731", "1 213", "1 321", "1 867", "2 123"),
        ("v1", "V1 fog Cornell box, 512x512, 256 spp, 16 b, **volumetric tracer**", "804-816", "807", "918", "951", "1 374"), ("v2", "V2 media_a, 512x512, 256 spp, 8 b, S=2, **volumetric tracer**", "528-531", "522", "531", "545", "1 039"),
        ("v3", "V3 volgrid_a (RGB grid volume), 512x512, 128 spp, 8 b, S=2, **volumetric tracer**", "543", "533", "515", "519", "1 074")]
out = []
for c, name, r1, r2, r3, r4, r5 in rows:
    d = json.load(open(f"{P}_bench_{c}.json")); r = d["roofline"]; st = r["stages"]; cb = d["cpu_baseline"]; pa = d["parity"]
    col = lambda k: (f"{st[k]['GB/s']:.0f} ({100 * st[k]['frac']:.1f} %; VALU {100 * st[k]['valu']['busy_frac']:.0f} %)" if k in st else "in k_shade")      # (rays traced in place: one kernel per bounce)
    out.append(f"| {name} | {cb['value']:.1f} on {cb['cores']}T ({cb['at_8_threads']['value']:.1f}) | {r1} | {r2} | {r3} | {r4} | {r5} | **{d['value']:.0f}** | {r['one_lane_Msamples/s']:.0f} | {col('extend')} | {col('shade')} | {col('shadow')} | "
               f"{r['pipeline_GB/s']:.0f} ({100 * r['pipeline_frac']:.1f} %) | {pa['relMSE']:.1e} | {100 * pa['frac_within_1e-3']:.2f} % |")
s = open("BASELINE.md").read().split("\n"); it = iter(out)
open("BASELINE.md", "w").write("\n".join(next(it) if re.match(r"\| (C[1-5]|V[1-3]) ", l) else l for l in s))
t = []
for c in ("c1", "c2", "c3", "c4", "c5"):
    d = json.load(open(f"{P}_bench_{c}.json")); st = d["roofline"]["stages"]; cj = json.load(open(f"{P}_{c}_counters.json"))["kernels"]
    for k in ("extend", "shade", "shadow"):
        if k not in st: continue
        v = st[k]; kn = cj[k]["kernels"][0]
        name = ("k_extend_flat" if "extend_flat" in kn else "k_extend_dyn" if "extend_dyn" in kn else "k_shadow_flat" if "shadow_flat" in kn else "k_shadow_dyn" if "shadow_dyn" in kn
                else "k_shade_group (classes)" if "shade_group" in kn else "k_shade_traced" if "shade_traced" in kn else "k_shade")
        t.append(f"| {c.upper()} | {name} | {v['GB/s']:.0f} ({100 * v['frac']:.1f} %) | {v['traffic_over_algorithmic']:.2f} | {100 * v['valu']['busy_frac']:.0f} % | {100 * cj[k]['valu_lane_utilisation']:.0f} % | {cj[k]['valu_insts_per_unit']:.1f} |")
s = open("DESIGN.md").read().split("\n")
first = [i for i, l in enumerate(s) if re.match(r"\| C[1-5] \| k_", l)]
s[first[0]:first[-1] + 1] = t
open("DESIGN.md", "w").write("\n".join(s))
for c, *_ in rows:
    d = json.load(open(f"{P}_bench_{c}.json")); print(c, d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["one_lane_Msamples/s"])
