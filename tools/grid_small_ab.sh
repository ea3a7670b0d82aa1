for g in 1 2 4; do for c in c5:32 c3:64 c4:64; do cfg=${c%%:*}; spp=${c##*:}
APT_GRID_SMALL=$g python bench.py --config $cfg --steps 1 --warmup 1 --spp $spp --no-cpu-baseline --no-exclusive-pass --lanes 1 > /tmp/ab.json 2>/tmp/ab.err
python - <<PY
import json
d = json.load(open("/tmp/ab.json")); pk = d["roofline"]["per_kernel"]
print("grid_small=$g $cfg 1-lane", d["value"], {k: v["ms"] for k, v in pk.items() if k in ("extend", "shadow", "shade")})
PY
APT_GRID_SMALL=$g python bench.py --config $cfg --steps 1 --warmup 1 --spp $((spp*3)) --no-cpu-baseline --no-exclusive-pass --no-profile | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   3 lanes', d['value'])"
done; done
