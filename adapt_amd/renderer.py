"""`Renderer` — drop-in for AdaPT's `pt` renderer class, backed by the HIP library.

Mirrors the contract `render.py` drives in the reference (reference
`renderer/vanilla_renderer.py:23-124`, `tracer/path_tracer.py:181-211`,
`tracer/tracer_base.py:36-102`):

    rdr = Renderer(emitters, array_info, objects, prop)      # same four values scene_parsing returns
    rdr.render(t_start, t_end, s_start, s_end, max_bnc, max_depth)   # +1 spp; the six ints are ignored, as upstream
    rdr.pixels.to_numpy()          # (w, h, 3) float32, index [x, y], linear radiance = color / cnt
    rdr.cnt[None], rdr.color.to_numpy(), rdr.w, rdr.h, rdr.do_crop, rdr.start_x ... rdr.end_y
    rdr.get_check_point() / rdr.load_check_point(d), rdr.reset(), rdr.summary()

Extensions (keyword-only, all optional): `n_spp=` on render() to queue many samples per
call (the wavefront batches them), `device/rank/world_size/band_width` for image-tile
sharding across GPUs, `seed`, `spp_per_batch`, `profile`, and film/bounce overrides so the
BASELINE configs can be run from one scene file.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import List, Optional

import numpy as np

from . import _lib
from .scene_pack import FlatScene, RenderConfig, make_config, pack_scene
from .tiles import TilePlan

__all__ = ["Renderer", "VolumeRenderer", "DeviceScene", "bxdf_probe", "medium_probe", "rng_stream"]


def _fp(a):
    return a.ctypes.data_as(_lib.f32p)


def _ip(a):
    return a.ctypes.data_as(_lib.i32p)


def bxdf_probe(bxdf_i, bxdf_f, dirs12, world_ior: float = 1.0, sample: bool = False, seed: int = 0, device: int = 0) -> np.ndarray:
    """Run the device surface models on explicit inputs (apt_bxdf_probe): eval+pdf -> (n,4), sample -> (n,9)."""
    lib = _lib.load()
    bi = np.ascontiguousarray(bxdf_i, np.int32).reshape(-1, 4)
    bf = np.ascontiguousarray(bxdf_f, np.float32).reshape(-1, 13)
    dd = np.ascontiguousarray(dirs12, np.float32).reshape(-1, 12)
    n = bi.shape[0]
    out = np.zeros((n, 9 if sample else 4), np.float32)
    _lib.check(lib.apt_bxdf_probe(int(device), n, _ip(bi), _fp(bf), _fp(dd), float(world_ior), int(bool(sample)), int(seed) & 0xffffffff, _fp(out)),
               "apt_bxdf_probe")
    return out


def medium_probe(med_i, med_f, mode: int, in7, seed: int = 0, device: int = 0) -> np.ndarray:
    """apt_medium_probe: per test a medium row (type, 16 floats) and 7 inputs -> (n, 8); mode 0 sample_mfp, 1 sample_new_rays, 2 eval + transmittance"""
    lib = _lib.load()
    mi = np.ascontiguousarray(med_i, np.int32).reshape(-1); mf = np.ascontiguousarray(med_f, np.float32).reshape(-1, 16)
    x = np.ascontiguousarray(in7, np.float32).reshape(-1, 7)
    out = np.zeros((x.shape[0], 8), np.float32)
    _lib.check(lib.apt_medium_probe(int(device), x.shape[0], _ip(mi), _fp(mf), int(mode), _fp(x), int(seed) & 0xffffffff, _fp(out)), "apt_medium_probe")
    return out


def rng_stream(pixel: int, seed: int, sample: int, n: int, device: int = 0) -> np.ndarray:
    lib = _lib.load()
    out = np.zeros(n, np.uint32)
    _lib.check(lib.apt_rng_stream(int(device), pixel & 0xffffffff, seed & 0xffffffff, sample & 0xffffffff, int(n),
                                  out.ctypes.data_as(_lib.u32p)), "apt_rng_stream")
    return out


class DeviceScene:
    """Scene arrays resident in HBM (apt_scene handle); shareable by several renderers on one device."""

    def __init__(self, fs: FlatScene, device: int = 0, lib=None):
        lib = lib or _lib.load()
        self.lib = lib                          # the build this scene lives in (adapt_amd/_lib.py: "fast" or "exact"); handles never cross builds
        self.fs, self.device = fs, device
        keep = [np.ascontiguousarray(a) for a in (fs.prims, fs.normals, fs.v_normals, fs.obj_info, fs.obj_aabb, fs.emitter_id,
                                                  fs.bxdf_i, fs.bxdf_f, fs.src_i, fs.src_f)]
        p, n, vn, oi, ab, ei, bi, bf, si, sf = keep
        desc = _lib.SceneDesc(fs.n_prims, fs.n_objects, fs.n_sources, int(fs.has_vertex_normal), _fp(p), _fp(n), _fp(vn), _ip(oi),
                              _fp(ab), _ip(ei), _ip(bi), _fp(bf), _ip(si), _fp(sf), float(fs.world_ior))
        if fs.has_textures:                     # image textures: uv coordinates, per-object records, one atlas per map
            tex = [np.ascontiguousarray(fs.uvs, np.float32), np.ascontiguousarray(fs.tex_i, np.int32), np.ascontiguousarray(fs.tex_f, np.float32)]
            keep += tex
            desc.uvs, desc.tex_i, desc.tex_f = _fp(tex[0]), _ip(tex[1]), _fp(tex[2])
            for m, img in enumerate(fs.atlas):
                if img is not None:
                    img = np.ascontiguousarray(img, np.float32); keep.append(img)
                    desc.atlas[m] = _fp(img); desc.atlas_h[m], desc.atlas_w[m] = int(img.shape[0]), int(img.shape[1])
        if fs.med_i is not None:                # participating media (read by the volumetric tracer only)
            med = [np.ascontiguousarray(fs.med_i, np.int32), np.ascontiguousarray(fs.med_f, np.float32)]
            keep += med
            desc.med_i, desc.med_f = _ip(med[0]), _fp(med[1])
        if fs.vol_i is not None:                # grid volume (read by the volumetric tracer only)
            vol = [np.ascontiguousarray(fs.vol_i, np.int32), np.ascontiguousarray(fs.vol_f, np.float32), np.ascontiguousarray(fs.vol_grid, np.float32)]
            keep += vol
            desc.vol_i, desc.vol_f, desc.vol_grid = _ip(vol[0]), _fp(vol[1]), _fp(vol[2])
        h = C.c_void_p()
        _lib.check(lib.apt_scene_create(C.byref(desc), int(device), C.byref(h)), "apt_scene_create", lib)
        self.handle = h

    def texture_query(self, maps, objs, uv):
        """Texture.query on the device (parity probe): maps 0 albedo / 1 normal / 2 bump, (n,2) coordinates -> (n,3)"""
        mo = np.ascontiguousarray(np.stack([np.int32(maps), np.int32(objs)], 1), np.int32)
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        out = np.zeros((uv.shape[0], 3), np.float32)
        _lib.check(self.lib.apt_texture_probe(self.handle, uv.shape[0], _ip(mo), _fp(uv), _fp(out)), "apt_texture_probe", self.lib)
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.apt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Counter:
    """`rdr.cnt[None]` (tracer_base.py:102)."""

    def __init__(self, owner): self._o = owner
    def __getitem__(self, _): return self._o._cnt
    def __setitem__(self, _, v): self._o._set_cnt(int(v))


class _FieldView:
    """`rdr.pixels` / `rdr.color`: objects with .to_numpy() -> (w, h, 3) float32 [x][y]."""

    def __init__(self, owner, normalised: bool): self._o, self._n = owner, normalised
    def to_numpy(self): return self._o._image(self._n)
    def from_numpy(self, arr): self._o._set_accum(np.asarray(arr, np.float32), self._o._cnt)
    @property
    def shape(self): return (self._o.w, self._o.h)


class Renderer:
    VOLUMETRIC = False          # subclass switch: VolumeRenderer renders with the reference's vpt semantics

    def __init__(self, emitters: List, array_info: dict, objects: List, prop: dict, *,
                 device: int = 0, rank: int = 0, world_size: int = 1, band_width: int = 4,
                 seed: int = 0, spp_per_batch: int = 0, profile: bool = False,
                 width: Optional[int] = None, height: Optional[int] = None,
                 max_bounce: Optional[int] = None, num_shadow_ray: Optional[int] = None, volumetric: Optional[bool] = None,
                 exact: Optional[bool] = None):
        # exact = True: the bit-parity build (the reference's float32 arithmetic operation for operation; debugging and the exact
        # parity tests), False: the fast build, None: whatever adapt_amd._lib currently hands out (fast unless APT_EXACT=1)
        self.lib = _lib.load(None if exact is None else ("exact" if exact else "fast"))
        self.arithmetic = _lib.arithmetic(self.lib)
        if volumetric is None:
            volumetric = self.VOLUMETRIC
        self.flat: FlatScene = pack_scene(emitters, array_info, objects, prop)
        self.rc: RenderConfig = make_config(prop, width=width, height=height, max_bounce=max_bounce,
                                            num_shadow_ray=num_shadow_ray, seed=seed, volumetric=bool(volumetric))
        self.volumetric = bool(volumetric)
        rc = self.rc
        # attributes the reference's callers read (watermark.py:23-30, render.py:129, path_tracer.py:181-193)
        self.w, self.h = rc.width, rc.height
        self.do_crop = rc.do_crop
        self.start_x, self.end_x, self.start_y, self.end_y = rc.start_x, rc.end_x, rc.start_y, rc.end_y
        self.crop_x, self.crop_y, self.crop_rx, self.crop_ry = rc.crop_x, rc.crop_y, rc.crop_rx, rc.crop_ry
        self.max_bounce, self.num_shadow_ray = rc.max_bounce, rc.num_shadow_ray
        self.use_rr, self.use_mis, self.anti_alias, self.stratified_sample = rc.use_rr, rc.use_mis, rc.anti_alias, rc.stratified
        self.focal, self.inv_focal = rc.focal, rc.inv_focal
        self.num_objects, self.num_prims, self.src_num = self.flat.n_objects, self.flat.n_prims, self.flat.n_sources
        self.cam_orient, self.cam_t, self.cam_r = rc.cam_orient, rc.cam_t, rc.cam_r
        self.device, self.rank, self.world_size = int(device), int(rank), int(world_size)
        self.plan = TilePlan(self.w, self.h, band_width if world_size > 1 else self.w, world_size)
        self._cnt = 0
        self._t0 = time.time()

        self.scene = DeviceScene(self.flat, self.device, self.lib)
        cfg = _lib.RenderCfg()
        for name in ("width", "height", "start_x", "end_x", "start_y", "end_y", "max_bounce", "num_shadow_ray", "rr_bounce_th"):
            setattr(cfg, name, int(getattr(rc, name)))
        for name in ("do_crop", "use_rr", "use_mis", "anti_alias", "stratified", "brdf_two_sides"):
            setattr(cfg, name, int(bool(getattr(rc, name))))
        cfg.rr_threshold = float(rc.rr_threshold)
        cfg.cam_r = (C.c_float * 9)(*np.float32(rc.cam_r).reshape(-1).tolist())
        cfg.cam_t = (C.c_float * 3)(*np.float32(rc.cam_t).tolist())
        cfg.inv_focal, cfg.half_w, cfg.half_h = float(rc.inv_focal), float(rc.half_w), float(rc.half_h)
        cfg.seed = int(rc.seed) & 0xffffffff
        cfg.band_width, cfg.rank, cfg.world_size = self.plan.band_width, self.rank, self.world_size
        cfg.spp_per_batch, cfg.device, cfg.profile = int(spp_per_batch), self.device, int(bool(profile))
        cfg.volumetric = int(self.volumetric)
        h = C.c_void_p()
        _lib.check(self.lib.apt_renderer_create(self.scene.handle, C.byref(cfg), C.byref(h)), "apt_renderer_create", self.lib)
        self.handle = h
        nc, hh = C.c_int32(0), C.c_int32(0)
        _lib.check(self.lib.apt_tile_shape(self.handle, C.byref(nc), C.byref(hh)), "apt_tile_shape", self.lib)
        self.n_cols = int(nc.value)
        assert self.n_cols == len(self.plan.columns(self.rank))
        self.cnt = _Counter(self)
        self.pixels = _FieldView(self, True)
        self.color = _FieldView(self, False)

    # ------------------------------------------------------------ rendering
    def render(self, _t_start: int = 0, _t_end: int = 0, _s_start: int = 0, _s_end: int = 0, _a: int = 0, _b: int = 0,
               *, n_spp: int = 1):
        """Accumulate `n_spp` more samples for every owned pixel (asynchronous; reads synchronise)."""
        _lib.check(self.lib.apt_render(self.handle, int(n_spp)), "apt_render", self.lib)
        self._cnt += int(n_spp)

    def synchronize(self):
        _lib.check(self.lib.apt_synchronize(self.handle), "apt_synchronize", self.lib)

    def reset(self):
        """No-op, exactly like the reference (`TracerBase.reset` is an empty kernel, tracer_base.py:284-286)."""

    def clear(self):
        """Zero the accumulation, the sample counter and the statistics."""
        _lib.check(self.lib.apt_reset(self.handle), "apt_reset", self.lib)
        self._cnt = 0

    # ------------------------------------------------------------- readback
    def tile_accum(self) -> np.ndarray:
        """This rank's accumulation tile, (n_cols, h, 3) float32."""
        out = np.empty((self.n_cols, self.h, 3), np.float32)
        c = C.c_int32(0)
        _lib.check(self.lib.apt_get_accum(self.handle, _fp(out), C.byref(c)), "apt_get_accum", self.lib)
        return out

    def tile_pixels(self) -> np.ndarray:
        out = np.empty((self.n_cols, self.h, 3), np.float32)
        _lib.check(self.lib.apt_read_pixels(self.handle, _fp(out)), "apt_read_pixels", self.lib)
        return out

    def _image(self, normalised: bool) -> np.ndarray:
        tile = self.tile_pixels() if normalised else self.tile_accum()
        if self.world_size == 1:
            return tile
        from .tiles import gather_image
        return gather_image(self, normalised)

    def _set_cnt(self, v: int):
        self._set_accum(self.tile_accum(), v)

    def _set_accum(self, arr: np.ndarray, cnt: int):
        arr = np.ascontiguousarray(arr, np.float32)
        if arr.shape != (self.n_cols, self.h, 3):
            if arr.shape == (self.w, self.h, 3):
                arr = np.ascontiguousarray(arr[self.plan.columns(self.rank)])
            else:
                raise ValueError(f"accumulation must be ({self.w},{self.h},3) or the tile ({self.n_cols},{self.h},3)")
        _lib.check(self.lib.apt_set_accum(self.handle, _fp(arr), int(cnt)), "apt_set_accum", self.lib)
        self._cnt = int(cnt)

    def device_accum_ptr(self) -> int:
        p, c = C.c_void_p(), C.c_int32(0)
        _lib.check(self.lib.apt_device_ptr(self.handle, C.byref(p), C.byref(c)), "apt_device_ptr", self.lib)
        return int(p.value)

    def stream_ptr(self) -> int:
        p = C.c_void_p()
        _lib.check(self.lib.apt_stream(self.handle, C.byref(p)), "apt_stream", self.lib)
        return int(p.value or 0)

    def stats(self) -> dict:
        st = _lib.Stats()
        _lib.check(self.lib.apt_get_stats(self.handle, C.byref(st)), "apt_get_stats", self.lib)
        return st.as_dict()

    # ----------------------------------------------------------- unit entry points
    def intersect(self, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = o.shape[0]
        prim, t, uv = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 2), np.float32)
        _lib.check(self.lib.apt_intersect(self.handle, n, _fp(o), _fp(d), _ip(prim), _fp(t), _fp(uv)), "apt_intersect", self.lib)
        return prim, t, uv

    def occluded(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        tmax = np.ascontiguousarray(tmax, np.float32).reshape(-1)
        occ = np.zeros(o.shape[0], np.int32)
        _lib.check(self.lib.apt_occluded(self.handle, o.shape[0], _fp(o), _fp(d), _fp(tmax), _ip(occ)), "apt_occluded", self.lib)
        return occ

    def emitter_probe(self, in11, seed: int = 0) -> np.ndarray:
        """apt_emitter_probe: rows (src index, hit_pos, normal, ray_d, min_depth) -> (n,12)."""
        x = np.ascontiguousarray(in11, np.float32).reshape(-1, 11)
        out = np.zeros((x.shape[0], 12), np.float32)
        _lib.check(self.lib.apt_emitter_probe(self.scene.handle, x.shape[0], _fp(x), int(seed) & 0xffffffff, _fp(out)), "apt_emitter_probe", self.lib)
        return out

    def info(self) -> dict:
        b, nq, lds, tm = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        qb = C.c_int64(0)
        name = C.c_char_p()
        _lib.check(self.lib.apt_renderer_info(self.handle, C.byref(b), C.byref(nq), C.byref(qb), C.byref(lds), C.byref(name), C.byref(tm)), "apt_renderer_info", self.lib)
        return {"spp_per_batch": b.value, "n_subqueues": nq.value, "queue_bytes": qb.value, "lds_bytes": lds.value,
                "shade_variant": name.value.decode() if name.value else "",
                "traversal": {0: "bvh", 1: "sweep", 2: "tile", 3: "flat"}.get(tm.value, str(tm.value)), "arithmetic": self.arithmetic}

    # ------------------------------------------------------------ checkpoint
    def get_check_point(self) -> dict:
        """Same keys as the reference's pickle (path_tracer.py:181-193)."""
        return {"w": self.w, "h": self.h, "crop_x": self.crop_x, "crop_y": self.crop_y, "crop_rx": self.crop_rx,
                "crop_ry": self.crop_ry, "focal": self.focal, "num_objects": self.num_objects, "num_prims": self.num_prims,
                "cam_orient": np.array(self.cam_orient), "src_num": self.src_num, "cam_t": np.array(self.cam_t),
                "accumulation": self.color.to_numpy(), "counter": self._cnt}

    def load_check_point(self, check_point: dict):
        for key, val in check_point.items():
            if key in ("accumulation", "counter"):
                continue
            if key in ("cam_t", "cam_orient"):
                ok = np.abs(np.asarray(val) - np.asarray(getattr(self, key))).max() < 1e-4
            else:
                ok = val == getattr(self, key)
            if not ok:
                raise ValueError(f"'{key}' from the checkpoint is different.")
        self._set_accum(np.asarray(check_point["accumulation"], np.float32), int(check_point["counter"]))

    def summary(self) -> str:
        self.synchronize()
        msg = f"{'VPT' if self.volumetric else 'PT'} SPP = {self._cnt}. Rendering time: {time.time() - self._t0:.3f} s"
        print(msg)
        return msg

    def close(self):
        if getattr(self, "handle", None):
            self.lib.apt_renderer_destroy(self.handle)
            self.handle = None
        if getattr(self, "scene", None):
            self.scene.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VolumeRenderer(Renderer):
    """Drop-in for the reference's `VolumeRenderer` (renderer/vpt.py:29-50,145-262; `--type vpt`, the reference's default): same
    constructor and surface as `Renderer`, volumetric path tracing in homogeneous media (world medium, media attached to BSDF
    objects, null surfaces, transmittance-tracked light samples) and in one grid volume (`<volume>`, adapt_amd/volumes.py)."""
    VOLUMETRIC = True
