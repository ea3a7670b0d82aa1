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
LIB = os.path.join(HERE, "libadapt_mi.so")
SOURCES = ["api.hip", "bvh_gpu.hip", "bvh_build.cpp", "bvh_linear.cpp", "bvh_wide.cpp"]
HEADERS = ["vec.hpp", "rng.hpp", "shading.hpp", "traverse.hpp", "stages.hpp", "bvh_build.hpp", os.path.join("..", "..", "include", "adapt_mi.h"), "volumetric.hpp"]
# -ffp-contract=off: the arithmetic written in csrc/ is the arithmetic executed (no FMA fusion), which is what
# lets the HIP path and the CPU oracle agree bit-for-bit on almost every path (DESIGN.md "float parity").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         # every queue/list append here is already aggregated per wave by hand (ballot + one atomic from lane 0);
         # LLVM's atomic optimizer would wrap that in a second aggregation and serialise independent atomics
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain is required to build adapt_amd)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, extra_flags=(), verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    cmd = [hipcc(), *FLAGS, *extra_flags, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
