mkdir -p gpurun_out/j8
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/j8/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/j8/pytest.log | tail -3
for v in main norsv; do
  if [ $v = norsv ]; then export ADAPT_MI_LIB=$PWD/build_exp/libadapt_mi_norsv.so; fi
  mkdir -p gpurun_out/j8/$v
  for c in "c3 128" "c4 64" "c5 32"; do set -- $c
    python bench.py --config $1 --steps 1 --warmup 1 --spp $2 --no-cpu-baseline > gpurun_out/j8/$v/$1.json 2> gpurun_out/j8/$v/$1.err
    python - <<PY
import json
d=json.load(open("gpurun_out/j8/$v/$1.json")); pk=d["roofline"]["per_kernel"]
print("$v $1", d["value"], "one-lane", d["roofline"].get("one_lane_Msamples/s"), {k: round(v["ms"],2) for k, v in pk.items()})
PY
  done
done
