"""rocprofv3 passes of tools/pmc_round.sh -> profiles/r0N_<config>_counters.json: per stage kernel the HBM-side bytes and the issued
VALU instructions PER QUEUE UNIT, which bench.py scales by the units its own run processed (`roofline.traffic`, `roofline.valu`).

    python tools/make_counters_json.py <config> <spp> <rocprof output dir> <round>

Every pass rendered 1 + spp + spp samples per pixel (kernel-load render, warm-up step, timed step) and the counters are summed over
all dispatches of the pass; the units per sample come from the bench line the stats pass printed.  FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 (KiB units); WRITE_SIZE as is.  The file carries the hash of
adapt_amd/csrc/*: bench.py refuses to attach it to other kernels."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha256  # noqa: E402

cfg, spp, out, rnd = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
line = [l for l in open(os.path.join(out, "stats.log")) if l.startswith("{")][-1]
b = json.loads(line)
pix = b["config"]["width"] * b["config"]["height"]
samples = (1 + 2 * spp) * pix
ps = b["per_sample"]
units = {"extend": ps["n_extend"] * samples, "shade": ps["n_extend"] * samples, "shadow": ps["n_shadow_traced"] * samples}
unit_name = {"extend": "queued ray", "shade": "queue entry", "shadow": "traced shadow ray"}


def stage(name):
    for key, pre in (("extend", ("k_extend",)), ("shadow", ("k_shadow", "k_vshadow")), ("shade", ("k_shade", "k_vshade", "k_vevent"))):
        if any(p in name for p in pre):
            return key
    return None


def first_db(sub):
    fs = glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(fs[0]) if fs else None


acc = defaultdict(lambda: defaultdict(float))
names = defaultdict(set)
for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    d = first_db(sub)
    if not d:
        continue
    q = ("select s.kernel_name, p.name, sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch k on k.event_id = e.event_id join rocpd_info_kernel_symbol s on k.kernel_id = s.id group by s.kernel_name, p.name")
    for name, cname, val in d.execute(q):
        st = stage(name)
        if st:
            acc[st][cname] += val
            names[st].add(name.split("(")[0])
time_ns = defaultdict(float)
calls = defaultdict(int)
d = first_db("stats")
if d:
    q = ("select s.kernel_name, count(*), sum(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name")
    for name, n, ns in d.execute(q):
        st = stage(name)
        if st:
            time_ns[st] += ns; calls[st] += n
kern = {}
for st, c in acc.items():
    if units[st] <= 0:
        continue
    hbm = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024 + c.get("WRITE_SIZE", 0.0) * 1024
    e = {"unit": unit_name[st], "units_in_profile": int(units[st]), "bytes_per_unit": round(hbm / units[st], 2),
         "fetch_bytes_per_unit_x2": round(2.0 * c.get("FETCH_SIZE", 0.0) * 1024 / units[st], 2), "write_bytes_per_unit": round(c.get("WRITE_SIZE", 0.0) * 1024 / units[st], 2),
         "kernels": sorted(names[st])}
    if c.get("SQ_INSTS_VALU"):
        e["valu_insts_per_unit"] = round(c["SQ_INSTS_VALU"] / units[st], 4)
        e["salu_insts_per_unit"] = round(c.get("SQ_INSTS_SALU", 0.0) / units[st], 4)
    if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
        e["valu_lane_utilisation"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4)
    if c.get("SQ_WAVE_CYCLES"):
        e["wait_any_frac_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4) if c.get("SQ_WAIT_ANY") else None
    if time_ns[st]:
        e["trace_total_ms"] = round(time_ns[st] / 1e6, 3); e["trace_calls"] = calls[st]
        e["trace_avg_launch_us"] = round(time_ns[st] / calls[st] / 1e3, 2)
        e["trace_algorithmic_GB/s"] = round({"extend": 40.0, "shade": None, "shadow": None}[st] * units[st] / time_ns[st], 1) if st == "extend" else None
    kern[st] = e
res = {"config": cfg, "round": rnd, "csrc_sha256": csrc_sha256(), "spp_per_pass": 1 + 2 * spp, "per_sample": ps, "kernels": kern,
       "source": f"profiles/r{rnd:02d}_{cfg}_rocprofv3.txt: rocprofv3 --pmc passes (SQ_*, FETCH_SIZE, WRITE_SIZE: separate passes, never combined with API tracing) over "
                 f"`bench.py --config {cfg} --steps 1 --warmup 1 --spp {spp} --lanes 1`; FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md), KiB units; "
                 "all kernels of a stage summed (class kernels, walk passes)"}
os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles"), exist_ok=True)
path = os.path.join(ROOT, "gpurun_out", "profiles", f"r{rnd:02d}_{cfg}_counters.json")
json.dump(res, open(path, "w"), indent=1)
print(path, json.dumps({k: {a: v.get(a) for a in ("bytes_per_unit", "valu_insts_per_unit", "valu_lane_utilisation", "trace_avg_launch_us")} for k, v in kern.items()}))
