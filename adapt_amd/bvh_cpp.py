"""`bvh_cpp` — drop-in for AdaPT's pybind11 module of the same name (tracer/bvh/bvh.cpp:274-312), backed by libadapt_mi.so.

The reference imports it lazily and falls back to brute force when the import fails (tracer/path_tracer.py:143-156):

    from bvh_cpp import bvh_build
    bvh_minmax, node_minmax, bvh_info, node_info = bvh_build(primitives, obj_info, world_min, world_max)

Same signature and return convention here: `primitives` float32 (N, 3, 3), `obj_info` int32 (2, n_obj) = primitive counts and
sphere flags (PathTracer.prepare_for_bvh, path_tracer.py:222-230), the world box; four FLAT arrays come back, which the caller
reshapes to (-1, 2, 3), (-1, 2, 3), (-1, 2), (-1, 3).  To use it from an AdaPT checkout, put a two-line `bvh_cpp.py` on its path
(`from adapt_amd.bvh_cpp import bvh_build`).  Host-only: no GPU is needed for this entry point.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

__all__ = ["bvh_build"]


def bvh_build(obj_array, obj_info, world_min, world_max):
    lib = _lib.load()
    prims = np.ascontiguousarray(obj_array, np.float32)
    if prims.ndim != 3 or prims.shape[1:] != (3, 3):
        raise ValueError("bvh_build: obj_array must have shape (N, 3, 3)")        # the pybind11 module reads out of bounds instead
    info = np.ascontiguousarray(obj_info, np.int32)
    if info.ndim != 2 or info.shape[0] != 2:
        raise ValueError("bvh_build: obj_info must have shape (2, n_obj): primitive counts, sphere flags")
    wmin, wmax = np.ascontiguousarray(world_min, np.float32).reshape(3), np.ascontiguousarray(world_max, np.float32).reshape(3)
    cnt, flag = np.ascontiguousarray(info[0]), np.ascontiguousarray(info[1])
    n = prims.shape[0]
    h = C.c_void_p()
    fp = lambda a: a.ctypes.data_as(_lib.f32p)
    ip = lambda a: a.ctypes.data_as(_lib.i32p)
    _lib.check(lib.apt_bvh_build_linear(fp(prims), n, ip(cnt), ip(flag), info.shape[1], fp(wmin), fp(wmax), C.byref(h)), "apt_bvh_build_linear")
    try:
        m, n2 = C.c_int32(0), C.c_int32(0)
        _lib.check(lib.apt_linear_bvh_counts(h, C.byref(m), C.byref(n2)), "apt_linear_bvh_counts")
        bvh_minmax, node_minmax = np.zeros(n * 6, np.float32), np.zeros(m.value * 6, np.float32)
        bvh_info, node_info = np.zeros(n * 2, np.int32), np.zeros(m.value * 3, np.int32)
        _lib.check(lib.apt_linear_bvh_export(h, fp(bvh_minmax), fp(node_minmax), ip(bvh_info), ip(node_info)), "apt_linear_bvh_export")
    finally:
        lib.apt_linear_bvh_free(h)
    return bvh_minmax, node_minmax, bvh_info, node_info
