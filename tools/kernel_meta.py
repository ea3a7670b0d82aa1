#!/usr/bin/env python3
"""Per-kernel register allocation read from the code objects inside a built library (the AMDGPU metadata note of every gfx950 bundle):

    python tools/kernel_meta.py [adapt_amd/libadapt_mi.so] [filter]

rocprofv3's `arch_vgpr_count` column reads HALF of what a wave64 kernel allocates on gfx950 (64 for the 128-VGPR shade kernel, 36 for the
72-VGPR walk), so occupancy computed from a kernel-trace table comes out twice the truth; tools/summarize_prof.py therefore prints the
allocation recorded in the code object itself: .vgpr_count + .agpr_count (one unified file of 512 registers per SIMD lane, allocated in
granules of 8), and the waves per SIMD that follow from it.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_meta(lib):
    """{mangled kernel name: {"vgpr", "agpr", "sgpr", "scratch", "lds", "alloc", "waves_per_simd"}} of every gfx950 code object in `lib`."""
    meta = {}
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fatbin")
        if subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", lib], capture_output=True).returncode != 0 or not os.path.exists(fb):
            return meta
        blob = open(fb, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for k, at in enumerate(starts):                 # one bundle per translation unit with device code
            part = os.path.join(td, f"bundle{k}")
            open(part, "wb").write(blob[at:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            co = os.path.join(td, f"k{k}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co):
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                def g(key):
                    m = re.search(r"\." + key + r":\s+(\d+)", blk)
                    return int(m.group(1)) if m else 0
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name:
                    continue
                agpr = int(blk.split("\n")[0].strip() or 0)
                vgpr = g("vgpr_count")
                alloc = max(8, -(-(vgpr + agpr) // 8) * 8)
                meta[name.group(1)] = {"vgpr": vgpr, "agpr": agpr, "sgpr": g("sgpr_count"), "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size"),
                                       "alloc": alloc, "waves_per_simd": min(8, 512 // alloc)}
    return meta


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "adapt_amd", "libadapt_mi.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    names = sorted(kernel_meta(lib).items())
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(n for n, _ in names), capture_output=True, text=True).stdout.split("\n")
    except Exception:
        dem = [n for n, _ in names]
    for (n, m), d in zip(names, dem):
        d = re.sub(r"\(DevScene.*", "", d)
        if flt in d:
            print(f"{d[:72]:72s} vgpr {m['vgpr']:4d} agpr {m['agpr']:3d} alloc {m['alloc']:4d} waves/SIMD {m['waves_per_simd']} sgpr {m['sgpr']:4d} scratch {m['scratch']:5d} lds {m['lds']}")
