"""Build libadapt_mi.so (HIP kernels + C-ABI) in-tree for gfx950.

    python -m adapt_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so stays next to this file (git-ignored, but it
travels with the gpurun snapshot) so that the library a test loads is visibly the in-tree one.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libadapt_mi.so")                 # default build: fast arithmetic (APT_FAST=1)
LIB_EXACT = os.path.join(HERE, "libadapt_mi_exact.so")     # bit-parity build: the reference's float32 arithmetic, operation for operation
SOURCES = ["api.hip", "bvh_gpu.hip", "bvh_build.cpp", "bvh_linear.cpp", "bvh_wide.cpp", "flat_build.cpp"]
HEADERS = ["vec.hpp", "rng.hpp", "shading.hpp", "traverse.hpp", "stages.hpp", "shade_stage.hpp", "unit_kernels.hpp", "bvh_build.hpp", os.path.join("..", "..", "include", "adapt_mi.h"), "volumetric.hpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         # every queue/list append here is already aggregated per wave by hand (ballot + one atomic from lane 0);
         # LLVM's atomic optimizer would wrap that in a second aggregation and serialise independent atomics
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
         # the SLP vectoriser pairs independent float operations into v_pk_* instructions - same arithmetic, element for element - and its
         # 64-bit register pairs cost every shading kernel 12-20 VGPRs (the four-wave class group 127 + spills -> 115, the walk 72 -> 64,
         # the all-models kernel one wave -> two) for issue slots these kernels are not short of; without it, same box, old -> new
         # Msamples/s: C2 4163 -> 4267, C3 1095 -> 1142, V1 1214 -> 1277, V3 965 -> 1057, nothing slower (profiles/NOTES.md, round 5).
         # The flat sweep's two-rays-per-lane packed FMAs are written as vector types by hand and stay.
         "-fno-slp-vectorize"]
# Two builds of the same sources (DESIGN.md "float parity policy"), both with -ffp-contract=off and IEEE division / sqrt; the exact build evaluates
# transcendentals in double and rounds once, the product build with OCML's float functions (below).  Apart from those calls the shading arithmetic written in csrc/ is the arithmetic executed in BOTH (measured: FMA
# contraction in the shading code buys nothing - k_shade is bound by its dependent loads and Philox - but turns exact zeros such as
# a*b - b*a into rounding residues, and upstream's NaN-slab quirk then stops firing: +0.3 % path vertices on scenes/test/features_a.xml):
#  exact: the small-scene intersectors are the reference's loop operation for operation; the HIP path and the CPU oracle agree bit for
#         bit on almost every path.  The checker's build: the bit-exact parity tests run on it.
#  fast : what ships and what bench.py measures (APT_FAST code paths): intersectors re-derived for speed inside SURVEY 8(d)'s stated
#         tolerances (precomputed-transform records, explicit FMAs, one reciprocal per test; t within 1e-5 relative, images within
#         1e-3 (1 + |x|) / relMSE <= 1e-4 of the oracle on the same random stream).  No -ffast-math: NaN / inf semantics are part of the result.
VARIANT_FLAGS = {
    "exact": ["-ffp-contract=off", "-DAPT_FAST=0"],
    # (round 6) the product build evaluates cos / sin / tan / pow with OCML's float versions (1-2 ulp) instead of in double rounded once: on the
    # five-wave kernels of rounds 5-6 that is C2 +4.9 %, C3 +8.6 %, C4 +3 %, C1 +5 % (same box), with the parity figures of the double version
    # (C2 every pixel 99.92 %, C3 99.73 % within 1e-3 (1 + x); profiles/NOTES.md round 6).  Divisions and square roots stay IEEE in both builds:
    # -fno-hip-fp32-correctly-rounded-divide-sqrt would be another +2.5 % / +7 % but reaches the reference-order intersectors too (C3: 98.0 %).
    # -DAPT_FAST_DIV=1 (round 6): the divisions and square roots of the NON-DELTA shading code - light sampling, MIS weights, lobe sampling, throughput,
    # roulette - as v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 (vec.hpp sdiv ...); mirrors, glass and every intersector keep IEEE.  Same box, on top of
    # the float transcendentals: C2 4 873 -> 5 001, C3 1 257 -> 1 336, C4 1 975 -> 2 016; C2 every pixel 99.92 %, C3 99.62 % (with the delta code fast too: 97.8 %).
    "fast": ["-ffp-contract=off", "-DAPT_FAST=1", "-DAPT_EXACT_MATH=0", "-DAPT_FAST_DIV=1"],
}
VARIANT_LIB = {"fast": LIB, "exact": LIB_EXACT}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain is required to build adapt_amd)")


def stale(variant: str = "fast") -> bool:
    lib = VARIANT_LIB[variant]
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS) or os.path.getmtime(os.path.abspath(__file__)) > t


def build(force: bool = False, extra_flags=(), verbose: bool = False, variants=("fast", "exact")) -> str:
    """Compile the stale variants (in parallel: one hipcc process each).  Returns the path of the default (fast) library."""
    procs = []
    for v in variants:
        if not force and not stale(v):
            continue
        cmd = [hipcc(), *FLAGS, *VARIANT_FLAGS[v], *extra_flags, *[os.path.join(CSRC, s) for s in SOURCES], "-o", VARIANT_LIB[v] + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        procs.append((v, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    failures = []
    for v, pr in procs:                      # every compile is waited for before anything is raised: no orphan hipcc, no half-written .tmp left behind
        out, err = pr.communicate()
        if pr.returncode != 0:
            failures.append(f"hipcc failed for the {v} build ({pr.returncode}):\n{out}\n{err}")
            if os.path.exists(VARIANT_LIB[v] + ".tmp"):
                os.remove(VARIANT_LIB[v] + ".tmp")
        else:
            os.replace(VARIANT_LIB[v] + ".tmp", VARIANT_LIB[v])
    if failures:
        raise RuntimeError("\n".join(failures))
    return LIB


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if a in VARIANT_LIB]
    print(build(force="--force" in sys.argv, verbose=True, variants=tuple(only) or ("fast", "exact")))
