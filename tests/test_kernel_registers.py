"""Occupancy guard for the hot kernels.  A kernel's register allocation decides how many waves a SIMD holds (512 VGPRs per lane: <= 128 ->
four waves, <= 96 -> five, <= 72 -> seven), and it is the maximum over every path the compiler sees, including ones a config never takes: an innocent
change elsewhere in the stage headers can cost C2's shade kernel its fourth wave (-10 %: it happened, profiles/NOTES.md).  This test compiles the
hot instantiations for gfx950 with -Rpass-analysis=kernel-resource-usage (hipcc cross-compiles without a GPU; ~10 s) and holds each to
the budget it ships with."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# (explicit instantiation, mangled-name prefix, VGPR budget, why)
KERNELS = [
    ("k_shade_traced_lean<0x002, 0x01>(DevScene, Params, Queues, Counters*, int, int)", "_Z19k_shade_traced_leanILi2ELi1EE", 72, "C1 / C2: rays traced in place, seven waves per SIMD"),
    ("k_shade_group<0x05, 5, 0x002, 0x504, 0x200, 0x801, 0, 0>(DevScene, Params, Queues, Counters*, GroupIn, int, int)", "_Z13k_shade_groupILi5ELi5ELi2ELi1284ELi512ELi2049ELi0ELi0EE", 96, "C4: the lean group of class kernels, five waves per SIMD, spot lights"),
    ("k_extend_dyn<1>(DevScene, Params, Queues, Counters*, int, const uint32_t*, LdsPlan)", "_Z12k_extend_dynILi1EE", 72, "C4 / C5: closest-hit walk, seven waves per SIMD"),
    ("k_shade_group<0x03, 5, 0x002, 0x504, 0x200, 0x801, 0, 0>(DevScene, Params, Queues, Counters*, GroupIn, int, int)", "_Z13k_shade_groupILi3ELi5ELi2ELi1284ELi512ELi2049ELi0ELi0EE", 96, "C3 / C5: the lean group of class kernels, five waves per SIMD (one 8-byte scratch slot)"),
    ("k_shade_group<0x03, 4, 0x001, 0x040, 0x080, 0x008, 0x010, 0x020>(DevScene, Params, Queues, Counters*, GroupIn, int, int)", "_Z13k_shade_groupILi3ELi4ELi1ELi64ELi128ELi8ELi16ELi32EE", 128, "C3 / C5: every other class kernel in one four-wave group"),
    ("k_vshade_ev_group<0x03, 0, 4, 0x002, 0x040, 0x504, 0x200>(DevScene, Params, Queues, Counters*, VGroupIn, int)", "_Z17k_vshade_ev_groupILi3ELi0ELi4ELi2ELi64ELi1284ELi512EE", 128, "V1 / V2: the four-wave group of surface events of the volumetric tracer"),
    ("k_extend_flat<1, 1>(DevScene, Params, Queues, Counters*, int, const uint32_t*, LdsPlan)", "_Z13k_extend_flatILi1ELi1EE", 96, "C3: hot flat sweep, five waves per SIMD"),
]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_hot_kernels_keep_their_register_budget(tmp_path):
    src = tmp_path / "probe.hip"
    src.write_text("#include <hip/hip_runtime.h>\n#include <algorithm>\n#include <cmath>\n#include <cstdio>\n#include <cstdlib>\n#include <cstring>\n"
                   f'#include "{ROOT}/include/adapt_mi.h"\n#include "{ROOT}/adapt_amd/csrc/bvh_build.hpp"\n#include "{ROOT}/adapt_amd/csrc/shade_stage.hpp"\n#include "{ROOT}/adapt_amd/csrc/volumetric.hpp"\n'
                   + "".join(f"template __global__ void {inst};\n" for inst, *_ in KERNELS))
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fno-slp-vectorize", "-ffp-contract=off",
           "-DAPT_FAST=1", "-DAPT_EXACT_MATH=0", "-DAPT_FAST_DIV=1", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", str(src), "-o", str(tmp_path / "probe.o")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    usage = {}
    for blk in out.stderr.split("Function Name: ")[1:]:
        name = blk.split("\n")[0].split(" [")[0].strip()
        v = re.search(r"VGPRs: (\d+)", blk); s = re.search(r"VGPRs Spill: (\d+)", blk); sc = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk)
        usage[name] = (int(v.group(1)), int(s.group(1)) if s else 0, int(sc.group(1)) if sc else 0)
    for inst, prefix, budget, why in KERNELS:
        hit = [(n, u) for n, u in usage.items() if n.startswith(prefix)]
        assert len(hit) == 1, (prefix, sorted(usage))
        vgprs, spilled, scratch = hit[0][1]
        # a scratch budget of 16 bytes per lane: the lean group with area lights parks ONE 8-byte constant there under its five-wave
        # bound; anything beyond that is a spill in a hot loop
        assert vgprs <= budget and spilled <= 1 and scratch <= 16, f"{inst}: {vgprs} VGPRs, {spilled} spilled, {scratch} B of scratch; budget {budget} ({why})"
