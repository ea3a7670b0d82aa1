mkdir -p gpurun_out/j2
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/j2/pytest_fused.log 2>&1; tail -15 gpurun_out/j2/pytest_fused.log
for f in 1 0; do
  APT_FUSED=$f python bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/j2/c2_fused$f.json 2> gpurun_out/j2/c2_fused$f.err
  APT_FUSED=$f python bench.py --config c1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/j2/c1_fused$f.json 2> gpurun_out/j2/c1_fused$f.err
done
python - <<'PY'
import json
for c in ("c2","c1"):
    for f in (1,0):
        try:
            d=json.load(open(f"gpurun_out/j2/{c}_fused{f}.json")); pk=d["roofline"]["per_kernel"]
            print(c,"fused",f,d["value"],"one-lane",d["roofline"].get("one_lane_Msamples/s"),{k:round(v["ms"],3) for k,v in pk.items()})
        except Exception as e:
            print(c,f,"failed",e,open(f"gpurun_out/j2/{c}_fused{f}.err").read()[-600:])
PY
ADAPT_MI_LIB=$PWD/build_exp/libadapt_mi_ufh.so APT_FUSED=0 timeout 600 python -m pytest tests/test_gpu_fast.py -m gpu -q -k "no_systematic" > gpurun_out/j2/bias_ufh.log 2>&1; tail -8 gpurun_out/j2/bias_ufh.log
