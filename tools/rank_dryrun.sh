#!/bin/bash
# What the driver launches on the 8-GPU node, on ONE GPU: N ranks of bench.py under torch.distributed.run, every rank on cuda:0, tiles gathered
# over gloo.  Throughput means nothing here (the ranks share one device); the gathered image must equal the single-rank image bit for bit.
#   tools/rank_dryrun.sh <config> "<rank counts>"      (GPU box; writes gpurun_out/rank_dryrun.log)
CFG=${1:-c2}; RANKS=${2:-"2 4 8"}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 1 --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive-pass --dump-image gpurun_out/rank1.npy > gpurun_out/rank1.json 2> gpurun_out/rank1.err
for n in $RANKS; do
  for mode in strong weak; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --config $CFG --steps 1 --warmup 0 \
        --no-cpu-baseline --no-exclusive-pass --no-profile --single-device --backend gloo --scaling $mode --dump-image gpurun_out/rank${n}_$mode.npy > gpurun_out/rank${n}_$mode.json 2> gpurun_out/rank${n}_$mode.err
    python - <<PY
import json, numpy as np
try:
    d = json.loads([l for l in open("gpurun_out/rank${n}_$mode.json") if l.startswith("{")][-1])
    a, b = np.load("gpurun_out/rank1.npy"), np.load("gpurun_out/rank${n}_$mode.npy")
    spp1 = json.loads([l for l in open("gpurun_out/rank1.json") if l.startswith("{")][-1])["config"]["spp_per_step"]
    same = np.array_equal(a, b, equal_nan=True) if "$mode" == "strong" else None
    fin = np.isfinite(a).all(axis=2) & np.isfinite(b).all(axis=2)
    an, bn = a[fin] / spp1, b[fin] / d["config"]["spp_per_step"]
    rel = float(np.mean((an - bn) ** 2 / (an ** 2 + 1e-2)))
    print(f"$CFG  {d['n_gpus']} ranks on one device, $mode: spp/step {d['config']['spp_per_step']}, tiling {d['config'].get('tiling')}, per-rank render ms {d.get('per_rank', {}).get('render_ms_per_step')}, "
          f"gathered image == single-rank image: {same}, relMSE of the normalised images {rel:.2e}")
except Exception as e:
    print("$CFG $n $mode FAILED", e, open("gpurun_out/rank${n}_$mode.err").read()[-400:])
PY
  done
done | tee gpurun_out/rank_dryrun.log
rm -f gpurun_out/rank*.npy
