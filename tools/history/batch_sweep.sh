#!/bin/bash
# samples-per-batch / lane-count sweep on one config (GPU box): tools/batch_sweep.sh <config> "<batches>" "<lanes>"
CFG=${1:-c2}; BATCHES=${2:-"32 64 128 256"}; LANES=${3:-"3"}
for l in $LANES; do for b in $BATCHES; do
  v=$(python bench.py --config $CFG --steps 2 --warmup 1 --spp-per-batch $b --lanes $l --no-cpu-baseline --no-exclusive-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config'].get('queue_MiB'))")
  echo "$CFG lanes $l batch $b: $v"
done; done
