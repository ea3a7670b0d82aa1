#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=..]...  -> build_exp/libadapt_mi_<name>.so (kernel tuning experiments on the FAST build; load with ADAPT_MI_LIB).
# EXACT=1 tools/build_variant.sh ... builds on the exact flags instead; SLP=-fslp-vectorize puts the SLP vectoriser back.
name=$1; shift
mkdir -p build_exp
if [ "${EXACT:-0}" = "1" ]; then base="-ffp-contract=off -DAPT_FAST=0"; else base="-ffp-contract=off -DAPT_FAST=1 -DAPT_EXACT_MATH=${EXACT_MATH:-0} -DAPT_FAST_DIV=${FAST_DIV:-1}"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-fast-math -fvisibility=hidden -mllvm -amdgpu-atomic-optimizer-strategy=None ${SLP:--fno-slp-vectorize} $base "$@" \
    adapt_amd/csrc/api.hip adapt_amd/csrc/bvh_gpu.hip adapt_amd/csrc/bvh_build.cpp adapt_amd/csrc/bvh_linear.cpp adapt_amd/csrc/bvh_wide.cpp adapt_amd/csrc/flat_build.cpp -o build_exp/libadapt_mi_$name.so 2>&1 | grep -v "warning\|^$" | head -5
