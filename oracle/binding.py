"""ctypes binding of oracle/libpt_oracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by
anything under adapt_amd/.  Takes the same FlatScene / RenderConfig the product's
C-ABI takes, so both sides are fed identical bytes.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)
f64p = C.POINTER(C.c_double)


class SceneDesc(C.Structure):
    _fields_ = [("n_prims", C.c_int), ("n_objects", C.c_int), ("n_sources", C.c_int), ("has_vertex_normal", C.c_int),
                ("prims", f32p), ("normals", f32p), ("v_normals", f32p), ("obj_info", i32p), ("obj_aabb", f32p),
                ("emitter_id", i32p), ("bxdf_i", i32p), ("bxdf_f", f32p), ("src_i", i32p), ("src_f", f32p),
                ("world_ior", C.c_float),
                ("uvs", f32p), ("tex_i", i32p), ("tex_f", f32p), ("atlas", f32p * 3), ("atlas_w", C.c_int * 3), ("atlas_h", C.c_int * 3),
                ("med_i", i32p), ("med_f", f32p), ("vol_i", i32p), ("vol_f", f32p), ("vol_grid", f32p)]


class Cfg(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int),
                ("do_crop", C.c_int), ("start_x", C.c_int), ("end_x", C.c_int), ("start_y", C.c_int), ("end_y", C.c_int),
                ("max_bounce", C.c_int), ("num_shadow_ray", C.c_int),
                ("use_rr", C.c_int), ("use_mis", C.c_int), ("anti_alias", C.c_int), ("stratified", C.c_int),
                ("brdf_two_sides", C.c_int), ("use_bvh", C.c_int), ("rr_bounce_th", C.c_int),
                ("rr_threshold", C.c_float), ("cam_r", C.c_float * 9), ("cam_t", C.c_float * 3),
                ("inv_focal", C.c_float), ("half_w", C.c_float), ("half_h", C.c_float), ("seed", C.c_uint32),
                ("volumetric", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("n_samples", C.c_longlong), ("n_shade", C.c_longlong), ("n_shadow", C.c_longlong),
                ("n_lit", C.c_longlong), ("n_draws", C.c_longlong), ("n_extend", C.c_longlong), ("n_track", C.c_longlong)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc is in the image)."""
    src = os.path.join(_HERE, "pt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_scene_create.restype = C.c_void_p
        _lib.orc_scene_create.argtypes = [C.POINTER(SceneDesc), f32p, C.c_int]
        _lib.orc_scene_destroy.argtypes = [C.c_void_p]
        _lib.orc_texture_query.argtypes = [C.c_void_p, C.c_int, i32p, f32p, f32p]
        _lib.orc_render.argtypes = [C.c_void_p, C.POINTER(Cfg), f32p, i32p, C.c_int, C.c_int, C.POINTER(Stats)]
        _lib.orc_fresnel_equation.restype = C.c_float
        _lib.orc_fresnel_equation.argtypes = [C.c_float] * 4
    return _lib


def _fp(a):
    return a.ctypes.data_as(f32p)


def _ip(a):
    return a.ctypes.data_as(i32p)


def _f3(v):
    return np.ascontiguousarray(v, np.float32)


def make_cfg(rc) -> Cfg:
    c = Cfg()
    for name in ("width", "height", "start_x", "end_x", "start_y", "end_y", "max_bounce", "num_shadow_ray", "rr_bounce_th"):
        setattr(c, name, int(getattr(rc, name)))
    for name in ("do_crop", "use_rr", "use_mis", "anti_alias", "stratified", "brdf_two_sides", "use_bvh"):
        setattr(c, name, int(bool(getattr(rc, name))))
    c.rr_threshold = float(rc.rr_threshold)
    c.cam_r = (C.c_float * 9)(*np.float32(rc.cam_r).reshape(-1).tolist())
    c.cam_t = (C.c_float * 3)(*np.float32(rc.cam_t).tolist())
    c.inv_focal, c.half_w, c.half_h = float(rc.inv_focal), float(rc.half_w), float(rc.half_h)
    c.seed = int(rc.seed) & 0xffffffff
    c.volumetric = int(bool(getattr(rc, "volumetric", False)))
    return c


class OracleScene:
    """Owns an oracle-side copy of a FlatScene (+ the reference-layout BVH if asked)."""

    def __init__(self, fs, cam_t=(0., 0., 0.), build_bvh: bool = False):
        L = lib()
        self._keep = [np.ascontiguousarray(a) for a in (fs.prims, fs.normals, fs.v_normals, fs.obj_info, fs.obj_aabb,
                                                        fs.emitter_id, fs.bxdf_i, fs.bxdf_f, fs.src_i, fs.src_f)]
        p, n, vn, oi, ab, ei, bi, bf, si, sf = self._keep
        d = SceneDesc(fs.n_prims, fs.n_objects, fs.n_sources, int(fs.has_vertex_normal),
                      _fp(p), _fp(n), _fp(vn), _ip(oi), _fp(ab), _ip(ei), _ip(bi), _fp(bf), _ip(si), _fp(sf),
                      float(fs.world_ior))
        if getattr(fs, "tex_i", None) is not None:                  # image textures (maps: albedo, normal, bump)
            tex = [np.ascontiguousarray(fs.uvs, np.float32), np.ascontiguousarray(fs.tex_i, np.int32), np.ascontiguousarray(fs.tex_f, np.float32)]
            self._keep += tex
            d.uvs, d.tex_i, d.tex_f = _fp(tex[0]), _ip(tex[1]), _fp(tex[2])
            for m, img in enumerate(fs.atlas):
                if img is not None:
                    img = np.ascontiguousarray(img, np.float32); self._keep.append(img)
                    d.atlas[m] = _fp(img); d.atlas_h[m], d.atlas_w[m] = int(img.shape[0]), int(img.shape[1])
        if getattr(fs, "med_i", None) is not None:                  # participating media (volumetric path tracer)
            med = [np.ascontiguousarray(fs.med_i, np.int32), np.ascontiguousarray(fs.med_f, np.float32)]
            self._keep += med
            d.med_i, d.med_f = _ip(med[0]), _fp(med[1])
        if getattr(fs, "vol_i", None) is not None:                  # grid volume (volumetric path tracer)
            vol = [np.ascontiguousarray(fs.vol_i, np.int32), np.ascontiguousarray(fs.vol_f, np.float32), np.ascontiguousarray(fs.vol_grid, np.float32)]
            self._keep += vol
            d.vol_i, d.vol_f, d.vol_grid = _ip(vol[0]), _fp(vol[1]), _fp(vol[2])
        ct = _f3(cam_t)
        self.handle = C.c_void_p(L.orc_scene_create(C.byref(d), _fp(ct), int(build_bvh)))
        self.fs = fs

    def __del__(self):
        try:
            if self.handle:
                lib().orc_scene_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def texture_query(self, maps, objs, uv):
        """Texture.query (bxdf/texture.py:111-139): maps 0 albedo / 1 normal / 2 bump, (n,2) coordinates -> (n,3)"""
        mo = np.ascontiguousarray(np.stack([np.int32(maps), np.int32(objs)], 1), np.int32)
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((uv.shape[0], 3), np.float32)
        if lib().orc_texture_query(self.handle, uv.shape[0], _ip(mo), _fp(uv), _fp(out)) != 0:
            raise ValueError("texture_query: no such texture")
        return out

    # ---- whole-image render (Renderer.render x n_spp)
    def render(self, rc, n_spp: int, accum=None, cnt: int = 0, threads: int = 0):
        cfg = make_cfg(rc)
        if accum is None:
            accum = np.zeros((rc.width, rc.height, 3), np.float32)
        c = C.c_int(cnt)
        st = Stats()
        lib().orc_render(self.handle, C.byref(cfg), _fp(accum), C.byref(c), int(n_spp), int(threads), C.byref(st))
        return accum, c.value, st.as_dict()

    def trace_sample(self, rc, i, j, cnt, script=None, max_events=64):
        cfg = make_cfg(rc)
        col = np.zeros(3, np.float32)
        ev = np.zeros((max_events, 18), np.float32)     # obj, prim, t, direct rgb, emit*w rgb, throughput rgb, next ray o, d
        ne, nd = C.c_int(0), C.c_int(0)
        if script is not None:
            sc = np.ascontiguousarray(script, np.float64)
            sp, sn = sc.ctypes.data_as(f64p), len(sc)
        else:
            sp, sn = None, 0
        lib().orc_trace_sample(self.handle, C.byref(cfg), int(i), int(j), int(cnt), sp, sn, _fp(col), _fp(ev), max_events,
                               C.byref(ne), C.byref(nd))
        return col, ev[:ne.value].copy(), nd.value

    def explain_non_finite(self, rc, a, b, n_spp: int, max_pixels: int = 16):
        """Pixels whose finiteness differs between a HIP render `a` and this oracle's render `b` of the same samples: for each, the samples
        whose throughput stops being finite at a path vertex - a sampled direction with pdf == 0 (a cosine-hemisphere draw of exactly 0,
        one in 2^24), after which the throughput is spec / 0: +inf if the rounding residue of n_s . out is positive, NaN (zeroed at the end,
        vanilla_renderer.py:119) if it is not.  That residue hangs on the last bits of the barycentrics the un-normalised vertex normal is
        interpolated with, which the product build's intersectors return to 1e-6 relative, not to the bit: a
        knife-edge the stated intersector tolerance cannot pin.  Returns (explained, findings): explained = every mismatching pixel has
        such a vertex among its samples."""
        bad = np.argwhere(np.isfinite(a).all(axis=2) != np.isfinite(b).all(axis=2))
        findings, explained = [], True
        for i, j in bad[:max_pixels]:
            hits = []
            for cnt in range(1, int(n_spp) + 1):
                col, ev, nd = self.trace_sample(rc, int(i), int(j), cnt)
                fin = np.isfinite(ev[:, 9:12]).all(axis=1) if len(ev) else np.zeros(0, bool)
                k = np.flatnonzero(~fin)
                if len(k) and k[0] > 0:
                    hits.append({"sample": cnt, "vertex": int(k[0]) - 1, "throughput_after": [float(x) for x in ev[k[0], 9:12]], "sample_colour_finite": bool(np.isfinite(col).all())})
            findings.append({"pixel": [int(i), int(j)], "zero_pdf_vertices": hits})
            explained = explained and len(hits) > 0
        return bool(explained and len(bad) <= max_pixels), findings

    def intersect(self, o, d, use_bvh=False):
        o, d = _f3(o).reshape(-1, 3), _f3(d).reshape(-1, 3)
        n = o.shape[0]
        obj, prim = np.zeros(n, np.int32), np.zeros(n, np.int32)
        t, uv, ns = np.zeros(n, np.float32), np.zeros((n, 2), np.float32), np.zeros((n, 3), np.float32)
        lib().orc_intersect_batch(self.handle, int(use_bvh), n, _fp(o), _fp(d), _ip(obj), _ip(prim), _fp(t), _fp(uv), _fp(ns))
        return obj, prim, t, uv, ns

    def occluded(self, o, d, tmax, use_bvh=False):
        o, d, tmax = _f3(o).reshape(-1, 3), _f3(d).reshape(-1, 3), _f3(tmax).reshape(-1)
        occ = np.zeros(o.shape[0], np.int32)
        lib().orc_occluded_batch(self.handle, int(use_bvh), o.shape[0], _fp(o), _fp(d), _fp(tmax), _ip(occ))
        return occ

    def bvh_arrays(self):
        nn, nb = C.c_int(0), C.c_int(0)
        lib().orc_bvh_counts(self.handle, C.byref(nn), C.byref(nb))
        node_mm, node_info = np.zeros((nn.value, 2, 3), np.float32), np.zeros((nn.value, 3), np.int32)
        bvh_mm, bvh_info = np.zeros((nb.value, 2, 3), np.float32), np.zeros((nb.value, 2), np.int32)
        lib().orc_bvh_export(self.handle, _fp(node_mm), _ip(node_info), _fp(bvh_mm), _ip(bvh_info))
        return bvh_mm, node_mm, bvh_info, node_info

    def pix2ray(self, rc, i, j, cnt, script):
        cfg = make_cfg(rc)
        sc = np.ascontiguousarray(script, np.float64)
        out = np.zeros(3, np.float32)
        lib().orc_pix2ray(self.handle, C.byref(cfg), int(i), int(j), int(cnt), sc.ctypes.data_as(f64p), len(sc), _fp(out))
        return out

    def src_sample_hit(self, src_idx, hit_pos, script=None, key=0, seed=0):
        """script given: scripted RNG; script None: Philox stream (key, seed), sample 1."""
        sp, sn = _script(script)
        pos, inten = np.zeros(3, np.float32), np.zeros(3, np.float32)
        pdf, nd = C.c_float(0), C.c_int(0)
        lib().orc_src_sample_hit(self.handle, int(src_idx), _fp(_f3(hit_pos)), sp, sn, C.c_uint32(key), C.c_uint32(seed),
                                 _fp(pos), _fp(inten), C.byref(pdf), C.byref(nd))
        return pos, inten, pdf.value, nd.value

    def src_eval(self, src_idx, inci_dir, normal, min_depth, ray_d):
        le = np.zeros(3, np.float32)
        pdf = C.c_float(0)
        lib().orc_src_eval(self.handle, int(src_idx), _fp(_f3(inci_dir)), _fp(_f3(normal)), C.c_float(min_depth),
                           _fp(_f3(ray_d)), _fp(le), C.byref(pdf))
        return le, pdf.value


def bxdf_eval_pdf(bi, bf, world_ior, n_s, n_g, incid, out):
    ev = np.zeros(3, np.float32)
    pdf = C.c_float(0)
    bi, bf = np.ascontiguousarray(bi, np.int32), np.ascontiguousarray(bf, np.float32)
    lib().orc_bxdf_eval_pdf(_ip(bi), _fp(bf), C.c_float(world_ior), _fp(_f3(n_s)), _fp(_f3(n_g)), _fp(_f3(incid)), _fp(_f3(out)),
                            _fp(ev), C.byref(pdf))
    return ev, pdf.value


def _script(script):
    if script is None:
        return None, 0
    sc = np.ascontiguousarray(script, np.float64)
    _script.keep = sc
    return sc.ctypes.data_as(f64p), len(sc)


def bxdf_sample(bi, bf, world_ior, n_s, n_g, incid, script=None, key=0, seed=0):
    """script given: scripted RNG; script None: Philox stream (key, seed), sample 1."""
    bi, bf = np.ascontiguousarray(bi, np.int32), np.ascontiguousarray(bf, np.float32)
    spt, sn = _script(script)
    d, s = np.zeros(3, np.float32), np.zeros(3, np.float32)
    pdf, sp, nd = C.c_float(0), C.c_int(0), C.c_int(0)
    lib().orc_bxdf_sample(_ip(bi), _fp(bf), C.c_float(world_ior), _fp(_f3(n_s)), _fp(_f3(n_g)), _fp(_f3(incid)),
                          spt, sn, C.c_uint32(key), C.c_uint32(seed), _fp(d), _fp(s), C.byref(pdf), C.byref(sp), C.byref(nd))
    return d, s, pdf.value, bool(sp.value), nd.value


def rotation_between(a, b):
    R = np.zeros(9, np.float32)
    lib().orc_rotation_between(_fp(_f3(a)), _fp(_f3(b)), _fp(R))
    return R.reshape(3, 3)


def fresnel_equation(n_in, n_out, cos_inc, cos_ref):
    return float(lib().orc_fresnel_equation(n_in, n_out, cos_inc, cos_ref))


def rng_stream(pixel, seed, sample, n):
    out = np.zeros(n, np.uint32)
    lib().orc_rng_stream(C.c_uint32(pixel), C.c_uint32(seed), C.c_uint32(sample), n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def num_threads():
    return int(lib().orc_num_threads())


def philox(counter4, key2):
    c = (C.c_uint32 * 4)(*[int(x) & 0xffffffff for x in counter4])
    k = (C.c_uint32 * 2)(*[int(x) & 0xffffffff for x in key2])
    out = (C.c_uint32 * 4)()
    lib().orc_philox(c, k, out)
    return [int(x) for x in out]


def medium_probe(med_type, med_f16, mode, vec, key=0, seed=0):
    """orc_medium_probe: mode 0 sample_mfp (vec = [max_depth]) -> 6 floats; 1 sample_new_rays (vec = incid) -> 8; 2 eval + transmittance
    (vec = incid, out, depth) -> 4"""
    L = lib()
    L.orc_medium_probe.argtypes = [C.c_int, f32p, C.c_int, f32p, C.c_uint32, C.c_uint32, f32p]
    L.orc_medium_probe.restype = None
    f = np.ascontiguousarray(med_f16, np.float32); x = np.ascontiguousarray(vec, np.float32)
    out = np.zeros((6, 8, 4)[mode], np.float32)
    L.orc_medium_probe(int(med_type), _fp(f), int(mode), _fp(x), C.c_uint32(key), C.c_uint32(seed), _fp(out))
    return out
