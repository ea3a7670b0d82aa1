"""C-ABI surface of libadapt_mi.so on a machine without a GPU: the library loads, exports every symbol
include/adapt_mi.h declares, the host-only entry points (BVH build) work, and the render path refuses to
run without a HIP device instead of silently falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu
from adapt_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "adapt_mi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(apt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/adapt_mi.h but not exported"
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    assert b"gfx950" in lib.apt_version()


def test_no_product_code_touches_the_oracle():
    """The oracle is test infrastructure: nothing under adapt_amd/ may import, link or open it."""
    for base, _, files in os.walk(os.path.join(ROOT, "adapt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(base, f), errors="ignore").read()
                for pat in (r"^\s*(from|import)\s+oracle", r"libpt_oracle", r"pt_oracle\.", r"oracle[/\\]", r"#include\s+\"[^\"]*oracle"):
                    assert not re.search(pat, text, flags=re.M), (pat, os.path.join(base, f))


def _scene_desc(fs):
    keep = [np.ascontiguousarray(a) for a in (fs.prims, fs.normals, fs.v_normals, fs.obj_info, fs.obj_aabb, fs.emitter_id,
                                              fs.bxdf_i, fs.bxdf_f, fs.src_i, fs.src_f)]
    p, n, vn, oi, ab, ei, bi, bf, si, sf = keep
    fp, ip = (lambda a: a.ctypes.data_as(_lib.f32p)), (lambda a: a.ctypes.data_as(_lib.i32p))
    return _lib.SceneDesc(fs.n_prims, fs.n_objects, fs.n_sources, 1, fp(p), fp(n), fp(vn), ip(oi), fp(ab), ip(ei), ip(bi), fp(bf),
                          ip(si), fp(sf), 1.0), keep


@pytest.mark.skipif(has_gpu(), reason="this is the no-device failure path")
def test_render_path_fails_loudly_without_a_device(flat):
    lib = _lib.load()
    desc, keep = _scene_desc(flat("cbox"))
    h = C.c_void_p()
    rc = lib.apt_scene_create(C.byref(desc), 0, C.byref(h))
    assert rc == -2 and b"no HIP device" in lib.apt_last_error()
    with pytest.raises(_lib.AptError):
        _lib.check(rc, "apt_scene_create")
    from adapt_amd import load_renderer
    with pytest.raises(_lib.AptError):
        load_renderer(os.path.join(ROOT, "scenes", "cbox"), "c2_cbox.xml")


def test_bad_arguments_are_rejected():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.apt_bvh_build(None, 0, None, 0, C.byref(h)) == -1
    assert lib.apt_scene_create(None, 0, C.byref(h)) == -1
    assert lib.apt_render(None, 1) == -1 and lib.apt_get_stats(None, None) == -1
    assert b"apt_get_stats" in lib.apt_last_error()


def build_bvh(fs):
    lib = _lib.load()
    prims, info = np.ascontiguousarray(fs.prims), np.ascontiguousarray(fs.obj_info)
    h = C.c_void_p()
    _lib.check(lib.apt_bvh_build(prims.ctypes.data_as(_lib.f32p), fs.n_prims, info.ctypes.data_as(_lib.i32p), fs.n_objects, C.byref(h)))
    nn, npr, dep = C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(lib.apt_bvh_counts(h, C.byref(nn), C.byref(npr), C.byref(dep)))
    nodes, order = np.zeros((nn.value, 16), np.float32), np.zeros(npr.value, np.int32)
    _lib.check(lib.apt_bvh_export(h, nodes.ctypes.data_as(_lib.f32p), order.ctypes.data_as(_lib.i32p)))
    lib.apt_bvh_free(h)
    return nodes, order, dep.value


def prim_bounds(fs, k, sphere):
    v = fs.prims[k]
    if sphere:
        return v[0] - v[1], v[0] + v[1]
    return v.min(axis=0), v.max(axis=0)


@pytest.mark.parametrize("tag", ["cbox", "balls_mono", "glass_box"])
def test_own_bvh_invariants(tag, flat):
    """Own builder (csrc/bvh_build.cpp): every primitive in exactly one leaf, child boxes bound their subtree."""
    fs = flat(tag)
    nodes, order, depth = build_bvh(fs)
    assert sorted(order.tolist()) == list(range(fs.n_prims))
    sphere = np.repeat(fs.obj_info[:, 2], fs.obj_info[:, 1]).astype(bool)
    links = nodes[:, 12:14].copy().view(np.int32)
    seen_nodes, seen_slots = set(), []

    def walk(link, lo, hi, d):
        assert d <= depth + 1
        if link < 0:
            code = ~int(link)
            first, count = code >> 4, code & 15
            assert 0 < count <= 4 or fs.n_prims == 0
            for s in range(first, first + count):
                plo, phi = prim_bounds(fs, order[s], sphere[order[s]])
                assert np.all(plo >= lo) and np.all(phi <= hi)
                seen_slots.append(s)
            return
        assert link not in seen_nodes
        seen_nodes.add(int(link))
        nd = nodes[link]
        for c, (a, b) in enumerate(((0, 3), (6, 9))):
            clo, chi = nd[a:a + 3], nd[b:b + 3]
            if lo is not None:
                assert np.all(clo >= lo - 1e-6) and np.all(chi <= hi + 1e-6)
            walk(int(links[link, c]), clo, chi, d + 1)

    nd0 = nodes[0]
    walk(0, None, None, 0)
    assert sorted(seen_slots) == list(range(fs.n_prims)) and len(seen_nodes) == nodes.shape[0]


def test_bvh_on_subdivided_mesh():
    """A few thousand triangles: depth stays logarithmic and the build is deterministic."""
    from adapt_amd.scene_pack import FlatScene
    n = 48
    xs, ys = np.meshgrid(np.linspace(0, 5, n + 1, dtype=np.float32), np.linspace(0, 5, n + 1, dtype=np.float32), indexing="ij")
    z = (0.3 * np.sin(xs) * np.cos(ys)).astype(np.float32)
    P = np.stack([xs, z, ys], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    tris = np.concatenate([np.stack([a, b, c], axis=-2).reshape(-1, 3, 3), np.stack([a, c, d], axis=-2).reshape(-1, 3, 3)]).astype(np.float32)
    fs = type("FS", (), {})()
    fs.prims, fs.n_prims, fs.obj_info, fs.n_objects = tris, tris.shape[0], np.int32([[0, tris.shape[0], 0]]), 1
    nodes, order, depth = build_bvh(fs)
    nodes2, order2, _ = build_bvh(fs)
    assert np.array_equal(nodes.view(np.int32), nodes2.view(np.int32)) and np.array_equal(order, order2)
    assert sorted(order.tolist()) == list(range(tris.shape[0]))
    assert depth <= 2.5 * np.log2(tris.shape[0]) and nodes.shape[0] < tris.shape[0]


def test_struct_layouts_match_the_header():
    """ctypes mirrors of the three C structs: same member order and count as include/adapt_mi.h (a silent mismatch would shift
    every later field, e.g. the media tables or the `volumetric` switch)."""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "adapt_mi.h")).read(), flags=re.S)
    def members(struct):
        body = re.search(r"typedef struct " + struct + r"\s*\{(.*?)\}\s*" + struct + ";", text, flags=re.S).group(1)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.sub(r"\[.*?\]", "", part.strip().split()[-1].lstrip("*")))
        return names
    assert members("apt_scene_desc") == [n for n, _ in _lib.SceneDesc._fields_]
    assert members("apt_render_cfg") == [n for n, _ in _lib.RenderCfg._fields_]
    assert members("apt_stats") == [n for n, _ in _lib.Stats._fields_]


def test_volume_renderer_class():
    from adapt_amd.renderer import VolumeRenderer, Renderer
    assert VolumeRenderer.VOLUMETRIC and not Renderer.VOLUMETRIC and issubclass(VolumeRenderer, Renderer)
