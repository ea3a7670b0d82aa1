mkdir -p gpurun_out/j3
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/j3/pytest.log 2>&1; tail -4 gpurun_out/j3/pytest.log
bash tools/bench_quick.sh j3/vote
ADAPT_MI_LIB=$PWD/build_exp/libadapt_mi_novote.so bash tools/bench_quick.sh j3/novote
