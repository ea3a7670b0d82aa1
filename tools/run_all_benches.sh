#!/bin/bash
# Full-size bench lines for BASELINE.md (one MI355X).  Usage: bash tools/run_all_benches.sh  (through gpurun)
mkdir -p gpurun_out/bench
python bench.py --config c2 --steps 3 --warmup 1 > gpurun_out/bench/c2.json 2> gpurun_out/bench/c2.err
python bench.py --config c1 --steps 10 --warmup 2 --cpu-seconds 8 > gpurun_out/bench/c1.json 2> gpurun_out/bench/c1.err
python bench.py --config c3 --steps 2 --warmup 1 --cpu-seconds 10 > gpurun_out/bench/c3.json 2> gpurun_out/bench/c3.err
python bench.py --config c4 --steps 1 --warmup 1 --spp 156 --cpu-seconds 10 > gpurun_out/bench/c4.json 2> gpurun_out/bench/c4.err
python bench.py --config c5 --steps 1 --warmup 1 --spp 216 --cpu-seconds 10 > gpurun_out/bench/c5.json 2> gpurun_out/bench/c5.err
python bench.py --config v1 --steps 2 --warmup 1 --cpu-seconds 10 > gpurun_out/bench/v1.json 2> gpurun_out/bench/v1.err
python bench.py --config v3 --steps 2 --warmup 1 --cpu-seconds 10 > gpurun_out/bench/v3.json 2> gpurun_out/bench/v3.err
python bench.py --config v2 --steps 2 --warmup 1 --cpu-seconds 10 > gpurun_out/bench/v2.json 2> gpurun_out/bench/v2.err
for c in c1 c2 c3 c4 c5 v1 v2 v3; do python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench/$c.json"))
    print("$c", d["value"], "Msamples/s; cpu", d.get("cpu_baseline", {}).get("value"), "on", d.get("cpu_baseline", {}).get("cores"), "threads; parity", {k: d.get("parity", {}).get(k) for k in ("relMSE", "frac_within_1e-3", "max_abs", "spp")}, "roofline", d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"], "pipeline", d["roofline"]["pipeline_GB/s"])
except Exception as e:
    print("$c failed", e, open("gpurun_out/bench/$c.err").read()[-300:])
PY
done
