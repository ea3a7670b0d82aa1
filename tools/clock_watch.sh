#!/bin/bash
# shader clock and power while a bench config renders (GPU box): tools/clock_watch.sh <config> [steps]
CFG=${1:-c2}; STEPS=${2:-100}
python bench.py --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-exclusive-pass > /tmp/cw.json 2>/dev/null &
PID=$!
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power" | sed -E 's/.*\(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/power \1/' | tr "\n" " "; echo
  sleep 0.2
done | sort | uniq -c | sort -k3,3n | tail -12
python -c "import json; d=json.loads(open('/tmp/cw.json').read().strip().splitlines()[-1]); print('$CFG', d['value'], 'Msamples/s; probe sclk', d['roofline']['sclk_mhz'])"
