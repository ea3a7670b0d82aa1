"""Flatten the parsed scene into the plain arrays the C-ABI scene description takes.

This is the host half of what the reference does in `TracerBase.__init__`
(tracer_base.py:36-102: film / crop / camera), `load_primitives`
(tracer_base.py:117-134) and `PathTracer.initialze` (path_tracer.py:245-274):
same inputs (emitters, array_info, objects, prop), but instead of filling Taichi
fields it produces contiguous float32/int32 arrays laid out as include/adapt_mi.h
documents.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
from scipy.spatial.transform import Rotation

__all__ = ["FlatScene", "RenderConfig", "pack_scene", "pack_source", "pack_bxdf", "make_config", "fov2focal", "np_rotation_between"]


def fov2focal(fov: float, img_size) -> float:
    """focal = 0.5 * size / tan(fov/2), fov in degrees (la/cam_transform.py:20-22)."""
    return 0.5 * img_size / np.tan(.5 * (fov / 180. * np.pi))


def np_rotation_between(fixed: np.ndarray, target: np.ndarray) -> np.ndarray:
    """Rotation taking `fixed` to `target` with the roll (z of 'zxy' euler) zeroed;
    +-I when (anti)parallel (la/cam_transform.py:31-49)."""
    axis = np.cross(fixed, target)
    dot = np.dot(fixed, target)
    if abs(dot) > 1. - 1e-5:
        return np.sign(dot) * np.eye(3, dtype=np.float32)
    axis /= np.linalg.norm(axis)
    axis *= np.arccos(dot)
    euler = Rotation.from_rotvec(axis).as_euler('zxy')
    euler[0] = 0
    return Rotation.from_euler('zxy', euler).as_matrix()


@dataclass
class FlatScene:
    prims: np.ndarray          # (N,3,3) f32   triangle vertices | (centre, rrr, 0)
    normals: np.ndarray        # (N,3)   f32   geometric normals
    v_normals: np.ndarray      # (N,3,3) f32   per-vertex shading normals (zeros when absent)
    obj_info: np.ndarray       # (n,3)   i32   first prim, prim count, 0 mesh | 1 sphere
    obj_aabb: np.ndarray       # (n,2,3) f32
    emitter_id: np.ndarray     # (n,)    i32   attached emitter or -1
    bxdf_i: np.ndarray         # (n,4)   i32   type, is_delta, is_bsdf, 0
    bxdf_f: np.ndarray         # (n,13)  f32   k_d k_s k_g mean ior
    src_i: np.ndarray          # (s,4)   i32   type, bool_bits, obj_ref_id, 0
    src_f: np.ndarray          # (s,11)  f32   intensity dir pos inv_area r
    has_vertex_normal: bool
    world_ior: float
    # image textures (None when the scene has none).  Maps: 0 albedo, 1 normal, 2 bump
    uvs: Optional[np.ndarray] = None        # (N,3,2) f32   per-vertex (u, v)
    tex_i: Optional[np.ndarray] = None      # (n,3,5) i32   type (-255 none), off_x, off_y, w, h
    tex_f: Optional[np.ndarray] = None      # (n,3,2) f32   scale_u, scale_v
    atlas: Optional[list] = None            # three (H,W,3) f32 images or None
    # participating media for the volumetric path tracer: row o < n = medium attached to object o's BSDF, row n = the world's
    med_i: Optional[np.ndarray] = None      # (n+1,)   i32   type: -1 transparent, 0 hg, 1 multi-hg, 2 rayleigh, 3 mie
    med_f: Optional[np.ndarray] = None      # (n+1,16) f32   ior, u_s, u_a, u_e, par, pdf
    # grid volume of the volumetric tracer (None when the scene declares none): adapt_amd/volumes.py
    vol_i: Optional[np.ndarray] = None      # (5,)     i32   type (2 = RGB), xres, yres, zres, phase type
    vol_f: Optional[np.ndarray] = None      # (33,)    f32   albedo, inv_T, trans, mini, maxi, majorant, majorant pdf, phase par, lobe weights
    vol_grid: Optional[np.ndarray] = None   # (z,y,x,3) f32  extinction per channel

    @property
    def has_textures(self): return self.tex_i is not None

    @property
    def has_scattering_media(self): return self.med_i is not None and bool((self.med_i >= 0).any())

    @property
    def has_volume(self): return self.vol_i is not None

    @property
    def n_prims(self): return int(self.prims.shape[0])
    @property
    def n_objects(self): return int(self.obj_info.shape[0])
    @property
    def n_sources(self): return int(self.src_i.shape[0])


@dataclass
class RenderConfig:
    width: int
    height: int
    do_crop: bool = False
    start_x: int = 0
    end_x: int = 0
    start_y: int = 0
    end_y: int = 0
    max_bounce: int = 8
    num_shadow_ray: int = 1
    use_rr: bool = True
    use_mis: bool = True
    anti_alias: bool = True
    stratified: bool = True
    brdf_two_sides: bool = False
    use_bvh: bool = False
    rr_bounce_th: int = 4
    rr_threshold: float = 0.1
    cam_r: np.ndarray = field(default_factory=lambda: np.eye(3, dtype=np.float32))
    cam_t: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    cam_orient: np.ndarray = field(default_factory=lambda: np.float32([0, 0, 1]))
    focal: float = 1.0
    inv_focal: float = 1.0
    half_w: float = 0.0
    half_h: float = 0.0
    seed: int = 0
    volumetric: bool = False      # False: Renderer (vanilla_renderer.py); True: VolumeRenderer (vpt.py)
    crop_x: int = 0
    crop_y: int = 0
    crop_rx: int = 0
    crop_ry: int = 0


_SRC_IDS = {"point": 0, "area": 1, "spot": 2, "collimated": 4}


def pack_source(e):
    """Emitter host object -> (int32[4] type, bool_bits, obj_ref_id, 0 ; float32[11] intensity dir pos inv_area r).
    Uses the object's own `pack()` when it has one (adapt_amd.emitters); otherwise reads the attributes of
    AdaPT's host classes (emitters/{point,area,spot,collimated}.py) so the reference's parser output works too."""
    if hasattr(e, "pack"):
        return e.pack()
    t = _SRC_IDS[e.type]
    fs = int(bool(getattr(e, "in_free_space", True))) << 4
    pos = np.float32(getattr(e, "pos", np.zeros(3))) if t != 1 else np.zeros(3, np.float32)
    direc = np.float32(getattr(e, "dir", (0, 0, 1))) if t in (2, 4) else np.float32([0, 0, 1])
    r = 0.0
    if t == 0:
        bits = 0x01 + fs
    elif t == 1:
        bits = fs | 0x04
    elif t == 2:
        bits, r = 0x01 + fs, float(e.half_cos)
    else:
        bits, r = int(e.radius == 0) + 0x02 + fs, float(e.radius)
    return (np.int32([t, bits, -1, 0]),
            np.concatenate([np.float32(e.intensity), direc, pos, [e.inv_area, r]]).astype(np.float32))


def pack_bxdf(b):
    """BRDF_np / BSDF_np host object -> (int32[4] type, is_delta, is_bsdf, 0 ; float32[13] k_d k_s k_g mean ior)."""
    if hasattr(b, "pack"):
        return b.pack()
    is_bsdf = hasattr(b, "medium")
    mean = np.float32([b.k_d.mean(), b.k_s.mean(), b.k_g.mean()])
    ior = np.float32(b.medium.ior) if is_bsdf else np.float32(1.0)
    return (np.int32([b.type_id, int(b.is_delta), int(is_bsdf), 0]),
            np.concatenate([b.k_d, b.k_s, b.k_g, mean, [ior]]).astype(np.float32))


def pack_medium(m):
    """Medium_np host object (adapt_amd.materials or AdaPT's bxdf/medium.py:24-68) -> (type, float32[16] ior u_s u_a u_e par pdf)."""
    if m is None:
        return -1, np.float32([1.0] + [0] * 12 + [1, 0, 0])
    if hasattr(m, "packed_medium"):                 # already flat (scenes rebuilt from fixtures)
        return m.packed_medium
    if not hasattr(m, "type_id"):                   # only an index of refraction is known
        return -1, np.float32([float(m.ior)] + [0] * 12 + [1, 0, 0])
    return int(m.type_id), np.concatenate([[m.ior], m.u_s, m.u_a, m.u_e, m.par, m.pdf]).astype(np.float32)


_TEX_MAPS = ("albedo", "normal", "bump")


def pack_texture(t):
    """Texture host object (adapt_amd.textures.Texture_np or AdaPT's bxdf/texture.py:33-94) -> (int32[5], float32[2])"""
    if t is None:
        return np.int32([-255, 0, 0, 0, 0]), np.float32([1, 1])
    if getattr(t, "type", None) == "checkerboard":
        raise NotImplementedError("checkerboard textures have no lookup upstream (bxdf/texture.py:96)")
    return np.int32([0, t.off_x, t.off_y, t.w, t.h]), np.float32([t.scale_u, t.scale_v])


def pack_scene(emitters: List, array_info: dict, objects: List, prop: dict) -> FlatScene:
    prims = np.ascontiguousarray(array_info["primitives"], dtype=np.float32)
    n_g = np.ascontiguousarray(array_info["n_g"], dtype=np.float32)
    n_s = np.ascontiguousarray(array_info["n_s"], dtype=np.float32)
    n_obj = len(objects)
    obj_info = np.zeros((n_obj, 3), np.int32)
    obj_aabb = np.zeros((n_obj, 2, 3), np.float32)
    emitter_id = np.full((n_obj,), -1, np.int32)
    bxdf_i = np.zeros((n_obj, 4), np.int32)
    bxdf_f = np.zeros((n_obj, 13), np.float32)
    src_i = np.zeros((len(emitters), 4), np.int32)
    src_f = np.zeros((len(emitters), 11), np.float32)
    for s, em in enumerate(emitters):
        src_i[s], src_f[s] = pack_source(em)
    first = 0
    for i, obj in enumerate(objects):
        obj_info[i] = (first, obj.tri_num, obj.type)
        first += obj.tri_num
        bxdf_i[i], bxdf_f[i] = pack_bxdf(obj.bsdf)
        obj_aabb[i] = obj.aabb
        emitter_id[i] = obj.emitter_ref_id
        if obj.emitter_ref_id >= 0:
            src_i[obj.emitter_ref_id, 2] = i           # obj_ref_id back-pointer (path_tracer.py:272-274)
    assert first == prims.shape[0]
    med_i = np.full((n_obj + 1,), -1, np.int32)
    med_f = np.zeros((n_obj + 1, 16), np.float32)
    for i, obj in enumerate(objects):
        med_i[i], med_f[i] = pack_medium(getattr(obj.bsdf, "medium", None))
    med_i[n_obj], med_f[n_obj] = pack_medium(prop["world"].medium)
    vol = {}
    if prop.get("volume"):
        from .volumes import GridVolume_np
        gv = prop["volume"][0]
        vi, vf, vg = (gv if isinstance(gv, GridVolume_np) else GridVolume_np(gv)).pack()
        vol = dict(vol_i=vi, vol_f=vf, vol_grid=vg)
    tex = {}
    images = prop.get("packed_textures")
    if images is not None and any(images.get(m) is not None for m in _TEX_MAPS):
        tex_i = np.zeros((n_obj, 3, 5), np.int32); tex_f = np.ones((n_obj, 3, 2), np.float32)
        for i, obj in enumerate(objects):
            group = getattr(obj, "texture_group", None) or {}
            for m, name in enumerate(_TEX_MAPS):
                tex_i[i, m], tex_f[i, m] = pack_texture(group.get(name))
                if tex_i[i, m, 0] > -255 and obj.type != 0:
                    raise NotImplementedError("textured spheres are not supported (upstream reads stale barycentrics there)")
        uvs = np.ascontiguousarray(array_info["uvs"], dtype=np.float32)
        tex = dict(uvs=uvs, tex_i=tex_i, tex_f=tex_f,
                   atlas=[None if images.get(m) is None else np.ascontiguousarray(images[m], np.float32) for m in _TEX_MAPS])
    return FlatScene(prims=prims, normals=n_g, v_normals=n_s, obj_info=obj_info, obj_aabb=obj_aabb,
                     emitter_id=emitter_id, bxdf_i=bxdf_i, bxdf_f=bxdf_f, src_i=src_i, src_f=src_f,
                     has_vertex_normal=bool(prop["has_vertex_normal"]),
                     world_ior=float(prop["world"].medium.ior), med_i=med_i, med_f=med_f, **vol, **tex)


def make_config(prop: dict, *, width: Optional[int] = None, height: Optional[int] = None,
                max_bounce: Optional[int] = None, num_shadow_ray: Optional[int] = None,
                seed: int = 0, use_bvh: Optional[bool] = None, volumetric: bool = False) -> RenderConfig:
    """Sensor dict -> RenderConfig.  Keyword overrides exist because the BASELINE
    configs differ from the values stored in the scene files."""
    film = prop["film"]
    w = int(width if width is not None else film["width"])
    h = int(height if height is not None else film["height"])
    crop_x, crop_y = film.get("crop_x", 0), film.get("crop_y", 0)
    crop_rx, crop_ry = film.get("crop_rx", 0), film.get("crop_ry", 0)
    do_crop = crop_rx > 0 and crop_ry > 0
    if do_crop:
        sx, ex, sy, ey = crop_x - crop_rx, crop_x + crop_rx, crop_y - crop_ry, crop_y + crop_ry
    else:
        sx, sy, ex, ey = 0, 0, w, h
    focal = fov2focal(prop["fov"], min(w, h))
    orient = np.array(prop["transform"][0], dtype=np.float32)
    orient = orient / np.linalg.norm(orient)
    cam_r = np.float32(np_rotation_between(np.float32([0, 0, 1]), orient))
    if use_bvh is None:
        use_bvh = prop.get("accelerator", "none") == "bvh"
    return RenderConfig(
        width=w, height=h, do_crop=do_crop, start_x=sx, end_x=ex, start_y=sy, end_y=ey,
        max_bounce=int(max_bounce if max_bounce is not None else prop["max_bounce"]),
        num_shadow_ray=int(num_shadow_ray if num_shadow_ray is not None else prop["num_shadow_ray"]),
        use_rr=bool(prop["use_rr"]), use_mis=bool(prop["use_mis"]), anti_alias=bool(prop["anti_alias"]),
        stratified=bool(prop["stratified_sampling"]), brdf_two_sides=bool(prop.get("brdf_two_sides", False)),
        use_bvh=bool(use_bvh), rr_bounce_th=int(prop.get("rr_bounce_th", 4)), rr_threshold=float(prop.get("rr_threshold", 0.1)),
        cam_r=np.ascontiguousarray(cam_r, np.float32), cam_t=np.float32(prop["transform"][1]), cam_orient=orient,
        focal=float(focal), inv_focal=float(1. / focal), half_w=w / 2, half_h=h / 2, seed=int(seed), volumetric=bool(volumetric),
        crop_x=crop_x, crop_y=crop_y, crop_rx=crop_rx, crop_ry=crop_ry)
