#!/bin/bash
# tools/lanes_ab.sh : render lanes 3 / 4 / 5 / 6 on the class-sorted configs (event-free timed region, no extra passes)
for l in 3 4 5 6; do for c in c3:384 c4:312 c5:324; do cfg=${c%%:*}; spp=${c##*:}
python bench.py --config $cfg --steps 1 --warmup 1 --spp $spp --lanes $l --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $l $cfg', d['value'], d['config']['queue_MiB'])"
done; done
