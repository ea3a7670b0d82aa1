#!/usr/bin/env python3
"""tools/resource_usage.py <remarks.txt> [filter]  -  per-kernel VGPR / SGPR / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for blk in txt.split("Function Name: ")[1:]:
    name = blk.split("\n")[0].split(" [")[0].strip()
    def g(k):
        m = re.search(k + r": (\d+)", blk); return int(m.group(1)) if m else -1
    rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize \[bytes/lane\]"), g("Occupancy \[waves/SIMD\]"), g("LDS Size \[bytes/block\]")))
try:
    dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
except Exception:
    dem = [r[0] for r in rows]
for r, d in zip(rows, dem):
    d = re.sub(r"\(DevScene.*", "", d)
    if flt in d:
        print(f"{d[:70]:70s} vgpr {r[1]:4d} agpr {r[2]:3d} sgpr {r[3]:4d} scratch {r[4]:5d} occ {r[5]:2d} lds {r[6]}")
