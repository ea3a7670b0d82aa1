#!/bin/bash
# quick A/B of the BVH configs on the GPU box: tools/bench_quick.sh <outdir> [extra env assignments...]
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
python bench.py --config c4 --steps 1 --warmup 1 --spp 128 --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err
python bench.py --config c5 --steps 1 --warmup 1 --spp 64 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err
for c in c4 c5; do python - <<PY
import json
try:
    d = json.load(open("$OUT/$c.json")); pk = d["roofline"]["per_kernel"]
    print("$c", d["value"], "one-lane", d["roofline"].get("one_lane_Msamples/s"), {k: v["ms"] for k, v in pk.items()})
except Exception as e:
    print("$c failed", e, open("$OUT/$c.err").read()[-400:])
PY
done
