"""HBM-side bytes per queue unit of the trace / shade kernels from a rocprofv3 PMC summary (tools/profile_c2.sh output) ->
profiles/r01_<config>_traffic.json, which bench.py scales by the units one launch processes (`roofline.traffic`).

    python tools/make_traffic_json.py <config> <spp of the profile run> <summary.txt> <bench json of the config>

The profile run is `bench.py --config <config> --steps 1 --warmup 1 --spp <spp>`: two renders.  FETCH_SIZE is taken x2-corrected as the
summary prints it (MI355X_MICROARCH.md), WRITE_SIZE as is; units per sample come from the bench line's `per_sample`."""
import json
import re
import sys

cfg, spp, summary, bench = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
b = json.load(open(bench))
pix = b["config"]["width"] * b["config"]["height"]
samples = 2 * spp * pix
ps = b["per_sample"]
units = {"extend": ps["n_extend"] * samples, "shade": ps["n_extend"] * samples, "shadow": ps["n_shadow_traced"] * samples}
text = open(summary).read()
tot = {"extend": 0.0, "shade": 0.0, "shadow": 0.0}
for m in re.finditer(r"^(k_\w+)[^\n]*\(dispatches \d+\)\n[^\n]*\n[^\n]*\n\s+HBM side: FETCH_SIZE [\d.]+ MiB raw, ([\d.]+) MiB x2-corrected;\s+WRITE_SIZE ([\d.]+) MiB", text, flags=re.M):
    name = m.group(1)
    key = "extend" if name.startswith("k_extend") else "shadow" if name.startswith(("k_shadow", "k_vshadow")) else "shade" if name.startswith(("k_shade", "k_vshade")) else None
    if key:
        tot[key] += (float(m.group(2)) + float(m.group(3))) * 2 ** 20
out = {"config": cfg, "kernels": {k: {"bytes_per_unit": round(tot[k] / units[k], 1), "unit": {"extend": "queued ray", "shade": "queue entry", "shadow": "shadow entry"}[k]}
                                  for k in tot if units[k] > 0 and tot[k] > 0},
       "source": f"{summary}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --config {cfg} --steps 1 --warmup 1 --spp {spp}`; "
                 "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for coalesced streams, KiB units; all kernels of a stage summed (class kernels, walk passes)"}
json.dump(out, open(f"profiles/r01_{cfg}_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"]))
