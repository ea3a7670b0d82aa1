#!/bin/bash
# tools/ru_build.sh <tag> [fast|exact] [-DFLAG...]  -> /tmp/ru/<tag>.txt : per-kernel register usage of api.hip (compile only, no link)
tag=$1; var=${2:-fast}; shift; shift
mkdir -p /tmp/ru
if [ "$var" = "exact" ]; then fl="-DAPT_FAST=0"; else fl="-DAPT_FAST=1"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fvisibility=hidden -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -ffp-contract=off $fl "$@" -Rpass-analysis=kernel-resource-usage -c adapt_amd/csrc/api.hip -o /tmp/ru/$tag.o > /tmp/ru/$tag.txt 2>&1
grep -E "error|Error" /tmp/ru/$tag.txt | head -5
python tools/resource_usage.py /tmp/ru/$tag.txt > /tmp/ru/$tag.tab
