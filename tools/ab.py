#!/usr/bin/env python3
"""A/B runs of bench.py on the GPU box, one line per run (through gpurun):

    python tools/ab.py <outdir> "<label>|<config>|<spp or 0>|<ENV=1 ENV2=x ...>|<extra bench flags>" ...

Prints: label, config, Msamples/s of the timed region (three lanes, no per-launch events), the one-lane rate and the per-kernel
milliseconds / launches of the exclusive one-lane pass.  ADAPT_MI_LIB=build_exp/libadapt_mi_<name>.so in the env column selects a
library built by tools/build_variant.sh."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", sys.argv[1])
os.makedirs(out, exist_ok=True)
for k, spec in enumerate(sys.argv[2:]):
    parts = (spec.split("|") + ["", "", "", ""])[:5]
    label, cfg, spp, envs, extra = [p.strip() for p in parts]
    env = dict(os.environ)
    for kv in envs.split():
        a, b = kv.split("=", 1)
        env[a] = b if not b.startswith("build_exp/") else os.path.join(root, b)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--config", cfg, "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + (["--spp", spp] if spp and spp != "0" else []) + extra.split()
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=root)
    tag = f"{k:02d}_{label}_{cfg}".replace(" ", "_").replace("/", "_")
    open(os.path.join(out, tag + ".err"), "w").write(res.stderr[-4000:])
    try:
        d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
        json.dump(d, open(os.path.join(out, tag + ".json"), "w"))
        pk = d["roofline"]["per_kernel"]
        print(f"{label:28s} {cfg} {d['value']:9.1f} Ms/s  one-lane {d['roofline'].get('one_lane_Msamples/s')}  " + "  ".join(f"{n} {v['ms']:.2f}/{v['launches']}" for n, v in pk.items()), flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{label:28s} {cfg} FAILED rc={res.returncode} {e}: {res.stderr[-600:]}", flush=True)
