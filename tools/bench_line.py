#!/usr/bin/env python3
"""Run bench.py with the given args and print a one-line digest (development aid)."""
import json
import os
import subprocess
import sys

tag = os.environ.get("TAG", "")
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", *sys.argv[1:]], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(tag, "FAILED", out.stderr[-400:])
    sys.exit(1)
d = json.loads(line[-1])
pk = d["roofline"]["per_kernel"]
print(tag, d["config"]["workload"][:16], d["value"], {k: round(v["ms"], 1) for k, v in pk.items()}, d["config"]["shade_variant"])
