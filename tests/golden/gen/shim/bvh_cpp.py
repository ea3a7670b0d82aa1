"""Stand-in for the reference's compiled `bvh_cpp` extension — GOLDEN GENERATION ONLY.

`tracer/bvh/bvh.cpp` needs Eigen and the (empty) ext/pybind11 submodule, so the real module cannot be built in this image.
This module gives `PathTracer.bvh_process` (tracer/path_tracer.py:143-179) what it imports: `bvh_build(obj_array, obj_info,
world_min, world_max)` with the return convention of bvh.cpp:274-296 — four flat arrays
(float[N*6] primitive boxes, float[M*6] node boxes, int[N*2] (object, primitive), int[M*3] (base, count, subtree size)) — so
that the reference's OWN traversal code (`convert_bvh_info`, `ray_intersect_bvh`, `does_intersect_bvh`, ti_bvh.py) runs unchanged
on a reference-layout tree.  The tree itself comes from the builder restated in oracle/pt_oracle.c (`orc_bvh_build_raw`), or,
with ADAPT_BVH_CPP=product, from the product's drop-in `apt_bvh_build_linear` (include/adapt_mi.h) — the fixture records
which.  What the fixtures pin is therefore the traversal on a given tree; the builder is checked by structure (tests/test_abi.py,
tests/test_oracle_properties.py).
"""
import ctypes as C
import os

import numpy as np

LAST = {}           # the arrays of the most recent build (the generator stores them in the fixture)


def bvh_build(obj_array, obj_info, world_min, world_max):
    prims = np.ascontiguousarray(obj_array, np.float32).reshape(-1, 9)
    info = np.ascontiguousarray(obj_info, np.int32)
    assert info.ndim == 2 and info.shape[0] == 2, "obj_info is (2, n_obj): primitive counts, sphere flags (path_tracer.py:222-230)"
    wmin, wmax = np.ascontiguousarray(world_min, np.float32), np.ascontiguousarray(world_max, np.float32)
    n, n_obj = prims.shape[0], info.shape[1]
    cnt, flag = np.ascontiguousarray(info[0]), np.ascontiguousarray(info[1])
    f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int)
    fp = lambda a: a.ctypes.data_as(f32p)
    ip = lambda a: a.ctypes.data_as(i32p)
    if os.environ.get("ADAPT_BVH_CPP", "oracle") == "product":
        from adapt_amd.bvh_cpp import bvh_build as product_build
        out = product_build(prims.reshape(-1, 3, 3), info, wmin, wmax)
        LAST.update(source="product", arrays=out)
        return out
    from oracle import binding as ob
    L = ob.lib()
    L.orc_bvh_build_raw.restype = C.c_int
    L.orc_bvh_build_raw.argtypes = [f32p, C.c_int, i32p, i32p, C.c_int, f32p, f32p, f32p, f32p, i32p, i32p]
    m = L.orc_bvh_build_raw(fp(prims), n, ip(cnt), ip(flag), n_obj, fp(wmin), fp(wmax), None, None, None, None)
    assert m > 0, "orc_bvh_build_raw failed"
    bvh_mm, node_mm = np.zeros(n * 6, np.float32), np.zeros(m * 6, np.float32)
    bvh_info, node_info = np.zeros(n * 2, np.int32), np.zeros(m * 3, np.int32)
    L.orc_bvh_build_raw(fp(prims), n, ip(cnt), ip(flag), n_obj, fp(wmin), fp(wmax), fp(bvh_mm), fp(node_mm), ip(bvh_info), ip(node_info))
    out = (bvh_mm, node_mm, bvh_info, node_info)
    LAST.update(source="oracle", arrays=out)
    return out
