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


@pytest.mark.parametrize("variant", ["fast", "exact"])
def test_library_exports_every_declared_symbol(variant):
    """both builds of the library (adapt_amd/build.py: the product and the bit-parity build) carry the whole C-ABI and say which one they are"""
    lib = _lib.load(variant)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/adapt_mi.h but not exported"
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    assert b"gfx950" in lib.apt_version() and f"arithmetic: {variant}".encode() in lib.apt_version()
    assert _lib.arithmetic(lib) == variant


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


# ---- the bvh_cpp.bvh_build drop-in (apt_bvh_build_linear, adapt_amd/bvh_cpp.py): reference layout, host only
def _world_box(fs, cam_t):
    """PathTracer.__init__, tracer/path_tracer.py:130-138: (objects U camera) -+ 0.1, seeded with +-1e3"""
    ab = np.asarray(fs.obj_aabb, np.float32).reshape(-1, 2, 3)
    lo = np.minimum(np.float32(cam_t), np.minimum(ab[:, 0].min(axis=0), np.float32(1e3)))
    hi = np.maximum(np.float32(cam_t), np.maximum(ab[:, 1].max(axis=0), np.float32(-1e3)))
    return np.float32(lo - np.float32(0.1)), np.float32(hi + np.float32(0.1))


def _linear(fs, cam_t):
    from adapt_amd.bvh_cpp import bvh_build
    info = np.stack([fs.obj_info[:, 1], fs.obj_info[:, 2]]).astype(np.int32)          # PathTracer.prepare_for_bvh: (2, n_obj)
    out = bvh_build(np.asarray(fs.prims, np.float32).reshape(-1, 3, 3), info, *_world_box(fs, cam_t))
    assert [a.ndim for a in out] == [1, 1, 1, 1] and [a.dtype for a in out] == [np.float32, np.float32, np.int32, np.int32]     # flat, as bvh.cpp:215-251 returns them
    return out[0].reshape(-1, 2, 3), out[1].reshape(-1, 2, 3), out[2].reshape(-1, 2), out[3].reshape(-1, 3)     # the reshapes of path_tracer.py:157-160


@pytest.mark.parametrize("tag", ["cbox", "balls_mono", "glass_box", "features_a"])
def test_bvh_cpp_drop_in_structure(tag, flat, parsed):
    """What AdaPT's stackless walk relies on (tracer/path_tracer.py:338-394): preorder, subtree-skip offsets, leaf <=> offset 1,
    leaves partition the primitives, boxes nest, (object, primitive) pairs consistent with the scene."""
    from adapt_amd.scene_pack import make_config
    fs = flat(tag)
    bvh_mm, node_mm, bvh_info, node_info = _linear(fs, make_config(parsed(tag)[3]).cam_t)
    M, N = node_info.shape[0], bvh_info.shape[0]
    assert N == fs.n_prims and tuple(node_info[0]) == (0, N, M)
    wlo, whi = _world_box(fs, make_config(parsed(tag)[3]).cam_t)
    assert np.array_equal(node_mm[0, 0], wlo) and np.array_equal(node_mm[0, 1], whi)         # the root carries the world box
    assert sorted(bvh_info[:, 1].tolist()) == list(range(N))
    leaf = node_info[:, 2] == 1
    assert node_info[leaf, 1].sum() == N and node_info[leaf, 1].min() >= 1
    for i in range(M):
        base, cnt, off = node_info[i]
        if off > 1:
            l = i + 1; r = l + node_info[l, 2]
            assert node_info[l, 2] + node_info[r, 2] + 1 == off and r + node_info[r, 2] == i + off
            assert node_info[l, 0] == base and node_info[r, 0] == base + node_info[l, 1] and node_info[l, 1] + node_info[r, 1] == cnt
            if i > 0:
                for c in (l, r):
                    assert np.all(node_mm[c, 0] >= node_mm[i, 0]) and np.all(node_mm[c, 1] <= node_mm[i, 1])
        else:
            assert np.all(bvh_mm[base:base + cnt, 0] >= node_mm[i, 0]) and np.all(bvh_mm[base:base + cnt, 1] <= node_mm[i, 1])
    obj_of_prim = np.repeat(np.arange(fs.n_objects), fs.obj_info[:, 1])
    assert np.array_equal(bvh_info[:, 0], obj_of_prim[bvh_info[:, 1]])
    # primitive boxes: vertex min / max, axes thinner than 1e-4 widened by 1e-4 (bvh_helper.h:30-45); spheres centre -+ radius
    sphere = np.repeat(fs.obj_info[:, 2], fs.obj_info[:, 1]).astype(bool)
    for slot in range(N):
        lo, hi = prim_bounds(fs, bvh_info[slot, 1], sphere[bvh_info[slot, 1]])
        if not sphere[bvh_info[slot, 1]]:
            thin = (hi - lo) < np.float32(1e-4)
            lo, hi = np.where(thin, lo - np.float32(1e-4), lo), np.where(thin, hi + np.float32(1e-4), hi)
        assert np.array_equal(bvh_mm[slot, 0], np.float32(lo)) and np.array_equal(bvh_mm[slot, 1], np.float32(hi))


def test_bvh_cpp_drop_in_known_answers(flat, parsed):
    """Node counts recorded from the reference's own bvh.cpp (probe (4)): the Cornell box's 34 primitives give 61 nodes
    (31 leaves, at most 2 primitives per leaf, root (0, 34, 61)); bunny.obj's 495 triangles give 975 nodes (488 leaves: 481 of one
    primitive, 7 of two).  Both depend on how libstdc++'s std::partition / std::nth_element order tied centroids."""
    import os
    from adapt_amd.bvh_cpp import bvh_build
    from adapt_amd.parsers.obj_loader import extract_obj_info
    from adapt_amd.scene_pack import make_config
    _, _, _, ni = _linear(flat("cbox"), make_config(parsed("cbox")[3]).cam_t)
    leaves = ni[ni[:, 2] == 1]
    assert tuple(ni[0]) == (0, 34, 61) and leaves.shape[0] == 31 and leaves[:, 1].max() == 2
    m, _, _, _ = extract_obj_info(os.path.join(ROOT, "scenes", "meshes", "cornell", "bunny.obj"))
    out = bvh_build(m, np.int32([[m.shape[0]], [0]]), m.min(axis=(0, 1)) - np.float32(0.1), m.max(axis=(0, 1)) + np.float32(0.1))
    ni = out[3].reshape(-1, 3)
    leaves = ni[ni[:, 2] == 1]
    assert ni.shape[0] == 975 and leaves.shape[0] == 488 and np.bincount(leaves[:, 1]).tolist() == [0, 481, 7]


def test_bvh_cpp_drop_in_matches_the_fixture_tree_and_rejects_bad_input(flat, parsed):
    """The tree the reference's own traversal walked when tests/golden/bvhref_cbox.npz was recorded (it came from the oracle's
    restated builder) is, array for array, the tree this entry point returns."""
    from adapt_amd.bvh_cpp import bvh_build
    from adapt_amd import _lib
    from conftest import golden, scene_from_golden
    from adapt_amd.scene_pack import make_config, pack_scene
    tup, g = scene_from_golden("cbox", prefix="bvhref")
    bvh_mm, node_mm, bvh_info, node_info = _linear(pack_scene(*tup), make_config(tup[3]).cam_t)
    assert np.array_equal(node_info, g["node_info"]) and np.array_equal(bvh_info, g["bvh_info"])
    assert np.array_equal(node_mm, g["node_minmax"]) and np.array_equal(bvh_mm, g["bvh_minmax"])
    tri = np.zeros((2, 3, 3), np.float32)
    with pytest.raises(_lib.AptError):
        bvh_build(tri, np.int32([[3], [0]]), np.zeros(3, np.float32), np.ones(3, np.float32))     # counts do not sum to N
    with pytest.raises(ValueError):
        bvh_build(tri.reshape(-1, 9), np.int32([[2], [0]]), np.zeros(3, np.float32), np.ones(3, np.float32))


@pytest.mark.parametrize("levels", [1, 3])
def test_bvh_cpp_drop_in_equals_the_oracle_builder_on_mesh_scenes(levels):
    """Two independent writings of the reference's builder - the product's C++ (real std::partition / std::nth_element) and the
    oracle's C (those algorithms restated) - return the same four arrays on the 5 950- and the 95 050-triangle scene."""
    from adapt_amd.scene_pack import make_config, pack_scene
    from adapt_amd.synth import three_bunnies
    from oracle import binding as ob
    tup = three_bunnies(levels)
    fs, rc = pack_scene(*tup), make_config(tup[3])
    mine = _linear(fs, rc.cam_t)
    theirs = ob.OracleScene(fs, rc.cam_t, build_bvh=True).bvh_arrays()
    for a, b in zip(mine, theirs):
        assert a.shape == b.shape and np.array_equal(a, b)
    ni = mine[3]
    assert ni[ni[:, 2] == 1, 1].max() <= 4          # a rejected split happens only in the <= 4 branch or when SAH says so


# ---- the 8-wide quantised tree the kernels walk (csrc/bvh_wide.cpp), exported through apt_bvh_wide_export
def build_wide(fs):
    lib = _lib.load()
    prims, info = np.ascontiguousarray(fs.prims), np.ascontiguousarray(fs.obj_info)
    h = C.c_void_p()
    _lib.check(lib.apt_bvh_build(prims.ctypes.data_as(_lib.f32p), fs.n_prims, info.ctypes.data_as(_lib.i32p), fs.n_objects, C.byref(h)))
    nn, lv = C.c_int32(), C.c_int32()
    _lib.check(lib.apt_bvh_wide_counts(h, C.byref(nn), C.byref(lv)))
    nodes, order = np.zeros((nn.value, 16), np.uint32), np.zeros(fs.n_prims, np.int32)
    _lib.check(lib.apt_bvh_wide_export(h, nodes.ctypes.data_as(_lib.u32p), order.ctypes.data_as(_lib.i32p)))
    gmin, gstep = np.zeros(3, np.float32), np.zeros(3, np.float32)
    _lib.check(lib.apt_bvh_wide_frame(h, gmin.ctypes.data_as(_lib.f32p), gstep.ctypes.data_as(_lib.f32p)))
    lib.apt_bvh_free(h)
    return nodes, order, lv.value, (gmin, gstep)


def decode_wide(nodes):
    """64-byte nodes (csrc/bvh_wide.cpp) -> per node: corner (3,) in grid units, scale (3,) = 2^e in grid units, leaf mask, inner mask,
    child_base, tri_base, qlo (3, 8), qhi (3, 8)"""
    w = nodes.astype(np.int64)
    corner = np.stack([w[:, 0] & 0xffff, w[:, 0] >> 16, w[:, 1] & 0xffff], 1)
    lmask, imask = (w[:, 1] >> 16) & 0xff, w[:, 1] >> 24
    ex = np.stack([(w[:, 2] >> 24) & 15, w[:, 2] >> 28, (w[:, 3] >> 24) & 15], 1)
    assert ((w[:, 3] >> 28) == 0).all()
    by = np.ascontiguousarray(nodes[:, 4:16]).view(np.uint8).reshape(-1, 6, 8)
    return corner, np.ldexp(1.0, ex), lmask, imask, w[:, 2] & 0xffffff, w[:, 3] & 0xffffff, by[:, 0:3], by[:, 3:6]


@pytest.mark.parametrize("scene", ["cbox", "balls_mono", "bunnies1"])
def test_wide_bvh_invariants(scene, flat):
    """Every primitive sits in exactly one leaf child (one primitive per leaf, at most eight per node); inner children are consecutive and
    counted by the inner-slot mask, leaf primitives by the leaf-slot mask; every decoded child box contains the boxes of all primitives
    below it (the quantisation rounds outwards, the 16-bit corner lies at or below the node's box)."""
    if scene == "bunnies1":
        from adapt_amd.scene_pack import pack_scene
        from adapt_amd.synth import three_bunnies
        fs = pack_scene(*three_bunnies(1))
    else:
        fs = flat(scene)
    nodes, order, levels, (gmin, gstep) = build_wide(fs)
    assert sorted(order.tolist()) == list(range(fs.n_prims))
    assert np.all(np.log2(gstep.astype(np.float64)) % 1 == 0)                           # power-of-two steps: the grid transform is exact
    corner, sc, lmask, imask, cbase, tbase, qlo, qhi = decode_wide(nodes)
    assert (corner >= 0).all() and (corner <= 65535).all() and (lmask & imask == 0).all()
    sphere = np.repeat(fs.obj_info[:, 2], fs.obj_info[:, 1]).astype(bool)
    seen_nodes, seen_slots, depth_seen = set(), [], [0]
    g0, gs = gmin.astype(np.float64), gstep.astype(np.float64)

    def walk(n, d):
        assert n not in seen_nodes and d <= levels
        seen_nodes.add(n); depth_seen[0] = max(depth_seen[0], d)
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        rank, tri_next = 0, 0
        for s in range(8):
            leaf, inner = bool((lmask[n] >> s) & 1), bool((imask[n] >> s) & 1)
            if not (leaf or inner):
                assert (qlo[n, :, s] == 255).all() and (qhi[n, :, s] == 0).all()          # inverted box: never hit
                continue
            blo = g0 + gs * (corner[n] + qlo[n, :, s].astype(np.float64) * sc[n])            # world = gmin + gstep * (corner + q * 2^e), exact in double
            bhi = g0 + gs * (corner[n] + qhi[n, :, s].astype(np.float64) * sc[n])
            if inner:
                clo, chi = walk(int(cbase[n]) + rank, d + 1)
                rank += 1
            else:
                slot = int(tbase[n]) + tri_next
                a, b = prim_bounds(fs, order[slot], sphere[order[slot]])
                clo, chi = a.astype(np.float64), b.astype(np.float64)
                seen_slots.append(slot)
                tri_next += 1
            assert np.all(blo <= clo) and np.all(bhi >= chi), (n, s)
            lo_all, hi_all = np.minimum(lo_all, clo), np.maximum(hi_all, chi)
        return lo_all, hi_all

    walk(0, 1)
    assert sorted(seen_slots) == list(range(fs.n_prims)) and len(seen_nodes) == nodes.shape[0] and depth_seen[0] == levels
    if fs.n_prims > 1000:
        assert nodes.shape[0] < fs.n_prims / 2.5 and levels <= 12         # 64-byte nodes over single-primitive leaves


def test_wide_bvh_walk_finds_every_brute_force_hit():
    """The device walk restated in numpy float32 (same fused slab arithmetic, same octant order, same stack discipline) on the
    5 950-triangle scene: the primitives it reaches always include the brute-force closest hit, for rays in every octant and for
    axis-parallel rays."""
    from adapt_amd.scene_pack import pack_scene
    from adapt_amd.synth import three_bunnies
    fs = pack_scene(*three_bunnies(1))
    nodes, order, levels, (gmin, gstep) = build_wide(fs)
    corner, sc, lmask, imask, cbase, tbase, qlo, qhi = decode_wide(nodes)
    ginv = (np.float32(1) / gstep).astype(np.float32)
    rs = np.random.RandomState(9)
    n = 300
    O = rs.uniform([0.3, 0.2, 0.3], [5.2, 5.2, 5.2], size=(n, 3)).astype(np.float32)
    D = rs.normal(size=(n, 3)).astype(np.float32); D /= np.linalg.norm(D, axis=1, keepdims=True)
    D[:12] = np.float32([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]] * 2)
    tri = fs.prims.astype(np.float64)
    e1, e2, p0 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0], tri[:, 0]

    def brute(o, d):                      # Moeller-Trumbore in double: which primitive is nearest
        pv = np.cross(d, e2); det = (e1 * pv).sum(1)
        with np.errstate(all="ignore"):
            inv = 1.0 / det; tv = o - p0; u = (tv * pv).sum(1) * inv; qv = np.cross(tv, e1); v = (qv * d).sum(1) * inv; t = (e2 * qv).sum(1) * inv
        ok = (np.abs(det) > 1e-12) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 1e-4)
        if not ok.any():
            return -1, np.inf
        k = np.argmin(np.where(ok, t, np.inf))
        return int(k), float(t[k])

    f32 = np.float32
    for o, d in zip(O, D):
        k_true, t_true = brute(o.astype(np.float64), d.astype(np.float64))
        with np.errstate(all="ignore"):
            inv = (np.where(np.abs(d) < 1e-30, np.copysign(f32(1e30), d), f32(1) / d).astype(np.float32) * gstep).astype(np.float32)      # the ray in grid units (traverse.hpp make_walk_ray)
        og = ((o - gmin).astype(np.float32) * ginv).astype(np.float32)
        noo = (-(og * inv)).astype(np.float32)
        octinv = 7 - ((4 if inv[0] < 0 else 0) | (2 if inv[1] < 0 else 0) | (1 if inv[2] < 0 else 0))
        reached, stack, tmax = set(), [(0, 0x80000000)], f32(1e7)
        while stack:
            base, bits = stack.pop()
            if bits <= 0x00ffffff:
                continue
            bit = bits.bit_length() - 1
            pim = bits & 0xff
            bits &= ~(1 << bit)
            if bits > 0x00ffffff:
                stack.append((base, bits))
            slot = (bit - 24) ^ octinv
            nidx = base + bin(pim & ((1 << slot) - 1)).count("1")
            s_ = (inv * sc[nidx].astype(np.float32)).astype(np.float32); c_ = (corner[nidx].astype(np.float32) * inv + noo).astype(np.float32)       # (numpy: separate roundings; the margin is the builder's padding)
            hit8 = 0
            for sl in range(8):
                if not ((int(lmask[nidx]) | int(imask[nidx])) >> sl) & 1:
                    continue
                qn = np.where(inv < 0, qhi[nidx, :, sl], qlo[nidx, :, sl]).astype(np.float32); qf = np.where(inv < 0, qlo[nidx, :, sl], qhi[nidx, :, sl]).astype(np.float32)
                tn = max(float((qn * s_ + c_).max()), 0.0); tf = min(float((qf * s_ + c_).min()), float(tmax))
                if not np.signbit(np.float32(tf) - np.float32(tn)):                    # the device's test: sign bit of (exit - entry)
                    hit8 |= 1 << sl
            for sl in range(8):
                if ((hit8 & int(lmask[nidx])) >> sl) & 1:
                    reached.add(int(order[int(tbase[nidx]) + bin(int(lmask[nidx]) & ((1 << sl) - 1)).count("1")]))
            hi_ = 0
            for sl in range(8):
                if ((hit8 & int(imask[nidx])) >> sl) & 1:
                    hi_ |= 1 << (sl ^ octinv)
            if hi_:
                stack.append((int(cbase[nidx]), (hi_ << 24) | int(imask[nidx])))
        assert k_true < 0 or k_true in reached, (o, d, k_true, t_true)


def test_host_threads_build_the_same_tree(monkeypatch):
    """Big scenes build on host threads (csrc/bvh_build.cpp: subtrees of n/64 primitives to workers; csrc/bvh_wide.cpp: one level of the
    8-wide tree at a time, numbered by a prefix sum).  The result must not depend on the thread count: byte-identical nodes, the same
    primitive order, the same depth as with APT_HOST_THREADS=1 - on a scene large enough (71 k primitives) to take both threaded paths."""
    from adapt_amd.scene_pack import pack_scene
    from adapt_amd.synth import bunny_field
    fs = pack_scene(*bunny_field(levels=2))
    assert fs.n_prims >= 65536
    out = {}
    for threads in ("1", "2", "7", "32"):
        monkeypatch.setenv("APT_HOST_THREADS", threads)
        out[threads] = build_wide(fs)
    for threads in ("2", "7", "32"):
        assert np.array_equal(out["1"][0], out[threads][0]) and np.array_equal(out["1"][1], out[threads][1]) and out["1"][2] == out[threads][2], threads
        assert np.array_equal(out["1"][3][0], out[threads][3][0]) and np.array_equal(out["1"][3][1], out[threads][3][1])
    assert sorted(out["1"][1].tolist()) == list(range(fs.n_prims))
