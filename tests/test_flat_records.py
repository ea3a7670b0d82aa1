"""The flat sweep's record builder (csrc/flat_build.cpp, reached through apt_flat_records) checked on the CPU: the records are evaluated
here in numpy - the same formulas the kernels use (traverse.hpp planar_solve / flat_loop / flat_resolve), in double on the float32 record
values - and the hits compared with the oracle's brute-force intersector (the reference's loop, tracer_base.py:168-237) on the same rays.
What it pins without a device: the rows U, V, T of every record, which triangle pairs were merged (and that concave or folded pairs were
not), the far-edge functions of convex quads, the per-triangle barycentric maps, the coplanar groups."""
import ctypes as C

import numpy as np
import pytest

from adapt_amd import _lib


def flat_records(prims, obj_info):
    lib = _lib.load()
    prims = np.ascontiguousarray(prims, np.float32).reshape(-1, 9); obj_info = np.ascontiguousarray(obj_info, np.int32).reshape(-1, 3)
    counts = np.zeros(7, np.int32); ns, nt = C.c_int32(0), C.c_int32(0)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)); ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    _lib.check(lib.apt_flat_records(fp(prims), prims.shape[0], ip(obj_info), obj_info.shape[0], ip(counts), None, 0, None, 0, C.byref(ns), C.byref(nt)), "apt_flat_records", lib)
    stream, tab = np.zeros(ns.value, np.float32), np.zeros(nt.value, np.float32)
    _lib.check(lib.apt_flat_records(fp(prims), prims.shape[0], ip(obj_info), obj_info.shape[0], ip(counts), fp(stream), ns.value, fp(tab), nt.value, C.byref(ns), C.byref(nt)), "apt_flat_records", lib)
    return counts, stream, tab


def sweep_records(counts, stream, tab, o, d):
    """numpy restatement of flat_loop<false> + flat_resolve (closest hit per ray): -> prim, t, uv"""
    n = o.shape[0]
    o64, d64 = o.astype(np.float64), d.astype(np.float64)
    best_t = np.full(n, 1e7); best = np.full(n, -1); best_uv = np.zeros((n, 2))
    at, rec = 0, 0
    kinds = [("para", counts[0] + counts[1], 12), ("gquad", counts[2] + counts[3], 18), ("tri", counts[4] + counts[5], 12)]
    for kind, cnt, stride in kinds:
        for _ in range(cnt):
            r = stream[at:at + stride].astype(np.float64)
            s = o64 - r[0:3]
            with np.errstate(all="ignore"):
                t = -(s @ r[9:12]) / (d64 @ r[9:12])
                u = s @ r[3:6] + t * (d64 @ r[3:6]); v = s @ r[6:9] + t * (d64 @ r[6:9])
                if kind == "para": inside = np.maximum(np.abs(u - 0.5), np.abs(v - 0.5)) <= 0.5
                elif kind == "tri": inside = np.minimum(np.minimum(u, v), 1.0 - u - v) >= 0.0
                else:
                    e1 = r[12] * u + r[13] * v + r[14]; e2 = r[15] * u + r[16] * v + r[17]
                    inside = np.minimum(np.minimum(u, v), np.minimum(e1, e2)) >= 0.0
                ok = inside & (t > 1e-4) & (t < best_t)
            best_t[ok] = t[ok]; best[ok] = rec; best_uv[ok, 0] = u[ok]; best_uv[ok, 1] = v[ok]
            at += stride; rec += 1
    for _ in range(counts[6]):                                  # spheres: the reference's formula (tracer_base.py:184-199), float32
        c, r2 = stream[at:at + 3], stream[at + 3]
        s2c = c[None, :] - o
        cn2 = (s2c * s2c).sum(1, dtype=np.float32); proj = (d * s2c).sum(1, dtype=np.float32)
        c2ray = cn2 - proj * proj
        with np.errstate(all="ignore"):
            cut = np.sqrt(r2 - c2ray)
            t = np.where(cn2 > r2 + np.float32(1e-4), proj - cut, proj + cut)
        ok = (c2ray < r2) & (t > 1e-4) & (t < best_t)
        best_t[ok] = t[ok]; best[ok] = rec; best_uv[ok] = 0.0
        at += 4; rec += 1
    ids = tab.reshape(-1, 28)[:, 8:12].copy().view(np.int32)
    maps = tab.reshape(-1, 28)[:, 12:24].astype(np.float64)
    hit = best >= 0
    b = np.maximum(best, 0)
    second = hit & (ids[b, 1] >= 0) & (best_uv.sum(1) > 1.0)
    prim = np.where(hit, np.where(second, ids[b, 1], ids[b, 0]), -1)
    m = np.where(second[:, None], maps[b, 6:12], maps[b, 0:6])
    uv = np.stack([m[:, 0] + m[:, 1] * best_uv[:, 0] + m[:, 2] * best_uv[:, 1], m[:, 3] + m[:, 4] * best_uv[:, 0] + m[:, 5] * best_uv[:, 1]], 1)
    return prim, np.where(hit, best_t, 1e7), uv


def rays(n, seed):
    rs = np.random.RandomState(seed)
    o = rs.uniform([0.1, 0.1, 0.1], [5.4, 5.3, 5.4], size=(n, 3)).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def check(counts, stream, tab, sc_intersect, is_sphere_prim, normals, seed):
    n_prims = normals.shape[0]
    o, d = rays(60000, seed)
    prim, t, uv = sweep_records(counts, stream, tab, o, d)
    obj_o, prim_o, t_o, uv_o, _ = sc_intersect(o, d)
    same = prim == prim_o
    hit = same & (prim_o >= 0)
    dt = np.abs(t - t_o.astype(np.float64))
    tri = hit & ~is_sphere_prim[np.maximum(prim_o, 0)]
    # t within 1e-5 relative - or, for grazing rays and rays that start next to the surface (where the float32 oracle's own
    # distance is that uncertain), the hit point within 5.5e-6 along the normal (tests/test_gpu_fast.py _check_hits)
    cos = np.abs(np.einsum("ij,ij->i", normals[prim_o[tri]].astype(np.float64), d[tri].astype(np.float64)))
    assert np.all((dt[tri] <= 1e-5 * np.abs(t_o[tri])) | (dt[tri] * cos <= 5.5e-6)), float((dt[tri] * cos).max())
    assert np.abs(uv[tri] - uv_o[tri]).max() <= 5e-5
    sph = hit & is_sphere_prim[np.maximum(prim_o, 0)]
    assert np.all(dt[sph] <= 2e-6 * np.abs(t_o[sph]) + 1e-6)       # (float32 numpy sums are not bit-equal to the C loop's; the GPU test is)
    diff = ~same                                                  # another primitive only at the same distance: shared edges, coplanar faces
    tied = np.abs(t[diff] - t_o[diff]) <= 1e-5 * np.maximum(np.abs(t_o[diff]), 1e-2)
    assert diff.mean() <= 5e-3 and (~tied).sum() <= 3, (int(diff.sum()), int((~tied).sum()))
    assert 2 * (counts[0] + counts[1] + counts[2] + counts[3]) + counts[4] + counts[5] + counts[6] == n_prims


@pytest.mark.parametrize("tag,expect", [("cbox", (10, 6, 2, 0)), ("balls_mono", (3, 2, 2, 6)), ("glass_box", None), ("features_a", None)])
def test_flat_records_reproduce_the_brute_force_hits(tag, expect, flat, oracle_scene):
    fs = flat(tag)
    counts, stream, tab = flat_records(fs.prims, fs.obj_info)
    sph = np.zeros(fs.prims.shape[0], bool)
    for first, cnt, is_s in fs.obj_info: sph[first:first + cnt] = bool(is_s)
    check(counts, stream, tab, oracle_scene(tag).intersect, sph, fs.normals, 11)
    if expect: assert (counts[0] + counts[1], counts[2] + counts[3], counts[4] + counts[5], counts[6]) == expect      # C2: 34 triangles -> 10 parallelograms + 6 convex quads + 2 triangles (one face of the measured Cornell blocks is not planar to 1e-6)
    if tag == "glass_box": assert counts[1] + counts[3] + counts[5] > 0                        # the box rests on the floor: a coplanar group


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_flat_records_merge_convex_pairs_only(seed):
    """Random quad soups (tests/test_gpu_fast.py _quad_soup): 6 convex quads + 2 pairs of a five-triangle fan -> 8 convex-quad records,
    3 parallelograms, and every concave, folded or edge-less pair left as triangles; hits against a brute force over the triangles."""
    from test_gpu_fast import _quad_soup
    from adapt_amd.scene_pack import pack_scene, make_config
    from oracle import binding as ob
    tup = _quad_soup(seed)
    fs = pack_scene(*tup)
    counts, stream, tab = flat_records(fs.prims, fs.obj_info)
    assert tuple(counts) == (3, 0, 8, 0, 23, 0, 0)
    sc = ob.OracleScene(fs, make_config(tup[3]).cam_t)
    check(counts, stream, tab, sc.intersect, np.zeros(fs.prims.shape[0], bool), fs.normals, 20 + seed)
