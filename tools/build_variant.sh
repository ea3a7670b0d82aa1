#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=..]...  -> build_exp/libadapt_mi_<name>.so (kernel tuning experiments; load with ADAPT_MI_LIB)
name=$1; shift
mkdir -p build_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fvisibility=hidden -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" \
    adapt_amd/csrc/api.hip adapt_amd/csrc/bvh_gpu.hip adapt_amd/csrc/bvh_build.cpp adapt_amd/csrc/bvh_linear.cpp adapt_amd/csrc/bvh_wide.cpp -o build_exp/libadapt_mi_$name.so 2>&1 | grep -v "warning\|^$" | head -5
