ADAPT_MI_LIB=$PWD/build_exp/libadapt_mi_h32.so bash tools/bench_quick.sh j6/h32
ADAPT_MI_LIB=$PWD/build_exp/libadapt_mi_h24.so bash tools/bench_quick.sh j6/h24
