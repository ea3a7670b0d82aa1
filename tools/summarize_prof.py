#!/usr/bin/env python3
"""Condense rocprofv3 output (rocpd sqlite: kernel trace + PMC passes) into one small text table per kernel.

    python tools/summarize_prof.py gpurun_out/prof > profiles/<name>.txt
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]


_DEM = {}


def short(name):
    """Demangled kernel name without its argument list (template arguments say which variant ran: k_shade / k_shade_traced<material mask,
    emitter mask, textures>, k_shade_group<emitter mask, waves, member masks...>, k_vshade_ev_group<emitter mask, grid volume, waves, member masks...>,
    k_extend / k_shadow / k_vshadow<traversal mode: 0 tree, 1 sweep, 2 tile>, k_extend_dyn / k_extend_flat<sorted, ...>)."""
    if name not in _DEM:
        import re
        import subprocess
        try:
            d = subprocess.run(["c++filt", name.replace(".kd", "")], capture_output=True, text=True).stdout.strip()
        except Exception:
            d = name
        d = re.sub(r"^void ", "", d)
        d = d.split("(")[0] if d.startswith("k_") else d.split("(")[0][:48]
        _DEM[name] = d
    return _DEM[name]


def first_db(sub):
    fs = glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(fs[0]) if fs else None


# Register columns: rocprofv3's arch_vgpr_count reads HALF of a wave64 kernel's allocation on gfx950 (64 for the 128-VGPR shade kernel), so
# the table carries what the code object itself records - .vgpr_count + .agpr_count in granules of 8 (tools/kernel_meta.py) - and the waves
# per SIMD that follow from it (512 registers per lane); `arch(rocprof)` is kept next to it so that the two can be told apart.
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_meta import kernel_meta  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
META = {}
for lib in (os.environ.get("ADAPT_MI_LIB"), os.path.join(ROOT, "adapt_amd", "libadapt_mi.so")):
    if lib and os.path.exists(lib):
        META = kernel_meta(lib)
        break

db = first_db("stats")
print("== kernel time (rocprofv3 --kernel-trace --stats), all dispatches of the run")
if db:
    q = ("select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
         "max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc")
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':40s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr_alloc':>10s} {'waves/SIMD':>10s} {'arch(rocprof)':>13s} {'sgpr':>5s} {'lds_B':>7s}")
    for r in rows:
        m = META.get(r[0].replace(".kd", ""))
        alloc, waves = (f"{m['alloc']:d}", f"{m['waves_per_simd']:d}") if m else ("?", "?")
        print(f"{short(r[0]):40s} {r[1]:6d} {r[2] / 1e3:11.1f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} {100 * r[2] / tot:6.2f} {alloc:>10s} {waves:>10s} {r[6]:13d} {r[7]:5d} {r[8]:7d}")

# Launch boundaries (round 6): the idle time between consecutive dispatches of the run - one render lane, so nothing else fills
# it - is the most a captured hipGraph (or any other way of taking the host off the path) could remove; with three lanes the other lanes'
# kernels run in these gaps anyway.  Only the last 90 % of the dispatches are counted (the head of the trace holds scene upload and warm-up).
if db:
    try:
        ev = list(db.execute("select d.start, d.end from rocpd_kernel_dispatch d order by d.start"))
        ev = ev[len(ev) // 10:]
        busy = sum(e - s for s, e in ev)
        gaps = [max(0, ev[i + 1][0] - ev[i][1]) for i in range(len(ev) - 1)]
        small = [g for g in gaps if g < 200_000]             # (gaps above 0.2 ms are step boundaries of the bench - host synchronisation - not launch gaps)
        print(f"\n== launch gaps, one lane: {len(small)} boundaries, {sum(small) / 1e3:.1f} us idle in total = {100 * sum(small) / max(1, busy + sum(small)):.2f} % of kernel time + gaps "
              f"(median {sorted(small)[len(small) // 2] / 1e3:.2f} us, mean {sum(small) / max(1, len(small)) / 1e3:.2f} us): the upper bound of what a captured hipGraph could save")
    except Exception as e:  # noqa: BLE001
        print(f"\n== launch gaps: not available ({e})")

print("\n== PMC passes (summed over the run's dispatches, per kernel)")
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for sub in ("pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_fetch", "pmc_write"):
    d = first_db(sub)
    if not d:
        continue
    q = ("select s.kernel_name, p.name, sum(e.value), count(distinct k.dispatch_id) from rocpd_pmc_event e "
         "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch k on k.event_id = e.event_id "
         "join rocpd_info_kernel_symbol s on k.kernel_id = s.id group by s.kernel_name, p.name")
    for name, cname, val, nd in d.execute(q):
        acc[short(name)][cname] += val
        disp[short(name)].add(nd)
for k, c in sorted(acc.items()):
    if not k.startswith("k_"):
        continue
    print(f"{k}  (dispatches {max(disp[k])})")
    print("    " + "  ".join(f"{n}={v:.5g}" for n, v in sorted(c.items())))
    w = c.get("SQ_WAVES")
    if w:
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, cycle constants)
        print(f"    per wave: VALU {c.get('SQ_INSTS_VALU', 0) / w:.0f}  SALU {c.get('SQ_INSTS_SALU', 0) / w:.0f}  SMEM {c.get('SQ_INSTS_SMEM', 0) / w:.0f}  "
              f"wave_cycles {4 * c.get('SQ_WAVE_CYCLES', 0) / w:.0f}  active_valu_cycles {4 * c.get('SQ_ACTIVE_INST_VALU', 0) / w:.0f}  "
              f"wait_inst_any_cycles {4 * c.get('SQ_WAIT_INST_ANY', 0) / w:.0f}")
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        # KiB units; FETCH_SIZE on gfx950 counts a wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM): x2 shown
        print(f"    HBM side: FETCH_SIZE {c.get('FETCH_SIZE', 0) / 1024:.1f} MiB raw, {2 * c.get('FETCH_SIZE', 0) / 1024:.1f} MiB x2-corrected;  "
              f"WRITE_SIZE {c.get('WRITE_SIZE', 0) / 1024:.1f} MiB")
