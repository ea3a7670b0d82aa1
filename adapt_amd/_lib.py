"""ctypes view of libadapt_mi.so (include/adapt_mi.h).  Loading fails loudly: there is no
Python or CPU fallback for the render path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Two builds of the same sources (adapt_amd/build.py): "fast" = the product (libadapt_mi.so: what render.py, bench.py and smoke() run),
# "exact" = the reference's float32 arithmetic operation for operation (libadapt_mi_exact.so: the build the bit-exact parity tests pin;
# APT_EXACT=1 makes it the default, `use("exact")` switches at run time).  Both export the same C-ABI; neither has a CPU fallback.
# ADAPT_MI_LIB: load another build in place of the fast one (kernel tuning experiments).
LIB_PATHS = {"fast": os.environ.get("ADAPT_MI_LIB") or os.path.join(_HERE, "libadapt_mi.so"),
             "exact": os.environ.get("ADAPT_MI_LIB_EXACT") or os.path.join(_HERE, "libadapt_mi_exact.so")}
_variant = "exact" if os.environ.get("APT_EXACT", "0") not in ("", "0") else "fast"
LIB_PATH = LIB_PATHS[_variant]

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
APT_N_KERNELS = 5
KERNEL_NAMES = ("generate", "extend", "shade", "shadow", "finalize")


class AptError(RuntimeError):
    pass


class SceneDesc(C.Structure):
    _fields_ = [("n_prims", C.c_int32), ("n_objects", C.c_int32), ("n_sources", C.c_int32), ("has_vertex_normal", C.c_int32),
                ("prims", f32p), ("normals", f32p), ("v_normals", f32p), ("obj_info", i32p), ("obj_aabb", f32p),
                ("emitter_id", i32p), ("bxdf_i", i32p), ("bxdf_f", f32p), ("src_i", i32p), ("src_f", f32p),
                ("world_ior", C.c_float),
                ("uvs", f32p), ("tex_i", i32p), ("tex_f", f32p), ("atlas", f32p * 3), ("atlas_w", C.c_int32 * 3), ("atlas_h", C.c_int32 * 3),
                ("med_i", i32p), ("med_f", f32p), ("vol_i", i32p), ("vol_f", f32p), ("vol_grid", f32p)]


class RenderCfg(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("do_crop", C.c_int32), ("start_x", C.c_int32), ("end_x", C.c_int32), ("start_y", C.c_int32), ("end_y", C.c_int32),
                ("max_bounce", C.c_int32), ("num_shadow_ray", C.c_int32),
                ("use_rr", C.c_int32), ("use_mis", C.c_int32), ("anti_alias", C.c_int32), ("stratified", C.c_int32),
                ("brdf_two_sides", C.c_int32), ("rr_bounce_th", C.c_int32), ("rr_threshold", C.c_float),
                ("cam_r", C.c_float * 9), ("cam_t", C.c_float * 3),
                ("inv_focal", C.c_float), ("half_w", C.c_float), ("half_h", C.c_float), ("seed", C.c_uint32),
                ("band_width", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
                ("spp_per_batch", C.c_int32), ("device", C.c_int32), ("profile", C.c_int32), ("volumetric", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("n_samples", C.c_int64), ("n_extend", C.c_int64), ("n_shade", C.c_int64), ("n_shadow", C.c_int64),
                ("n_shadow_traced", C.c_int64), ("n_lit", C.c_int64), ("n_draws", C.c_int64), ("n_poisoned", C.c_int64),
                ("launches", C.c_int64 * APT_N_KERNELS), ("kernel_ms", C.c_double * APT_N_KERNELS), ("render_ms", C.c_double),
                ("n_track", C.c_int64)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k in ("n_samples", "n_extend", "n_shade", "n_shadow", "n_shadow_traced", "n_lit", "n_draws", "n_poisoned", "n_track")}
        d["launches"] = dict(zip(KERNEL_NAMES, [int(x) for x in self.launches]))
        d["kernel_ms"] = dict(zip(KERNEL_NAMES, [float(x) for x in self.kernel_ms]))
        d["render_ms"] = float(self.render_ms)
        return d


# every symbol include/adapt_mi.h declares: (restype, argtypes)
SYMBOLS = {
    "apt_bvh_build": (C.c_int, [f32p, C.c_int32, i32p, C.c_int32, C.POINTER(C.c_void_p)]),
    "apt_bvh_counts": (C.c_int, [C.c_void_p, i32p, i32p, i32p]),
    "apt_bvh_export": (C.c_int, [C.c_void_p, f32p, i32p]),
    "apt_bvh_wide_counts": (C.c_int, [C.c_void_p, i32p, i32p]),
    "apt_flat_records": (C.c_int, [f32p, C.c_int32, i32p, C.c_int32, i32p, f32p, C.c_int32, f32p, C.c_int32, i32p, i32p]),
    "apt_bvh_wide_export": (C.c_int, [C.c_void_p, u32p, i32p]),
    "apt_bvh_wide_frame": (C.c_int, [C.c_void_p, f32p, f32p]),
    "apt_bvh_free": (None, [C.c_void_p]),
    "apt_bvh_build_linear": (C.c_int, [f32p, C.c_int32, i32p, i32p, C.c_int32, f32p, f32p, C.POINTER(C.c_void_p)]),
    "apt_linear_bvh_counts": (C.c_int, [C.c_void_p, i32p, i32p]),
    "apt_linear_bvh_export": (C.c_int, [C.c_void_p, f32p, f32p, i32p, i32p]),
    "apt_linear_bvh_free": (None, [C.c_void_p]),
    "apt_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int32, C.POINTER(C.c_void_p)]),
    "apt_scene_destroy": (None, [C.c_void_p]),
    "apt_renderer_create": (C.c_int, [C.c_void_p, C.POINTER(RenderCfg), C.POINTER(C.c_void_p)]),
    "apt_renderer_destroy": (None, [C.c_void_p]),
    "apt_render": (C.c_int, [C.c_void_p, C.c_int32]),
    "apt_synchronize": (C.c_int, [C.c_void_p]),
    "apt_tile_shape": (C.c_int, [C.c_void_p, i32p, i32p]),
    "apt_read_pixels": (C.c_int, [C.c_void_p, f32p]),
    "apt_get_accum": (C.c_int, [C.c_void_p, f32p, i32p]),
    "apt_set_accum": (C.c_int, [C.c_void_p, f32p, C.c_int32]),
    "apt_reset": (C.c_int, [C.c_void_p]),
    "apt_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "apt_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), i32p]),
    "apt_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "apt_intersect": (C.c_int, [C.c_void_p, C.c_int32, f32p, f32p, i32p, f32p, f32p]),
    "apt_occluded": (C.c_int, [C.c_void_p, C.c_int32, f32p, f32p, f32p, i32p]),
    "apt_rng_stream": (C.c_int, [C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, u32p]),
    "apt_bxdf_probe": (C.c_int, [C.c_int32, C.c_int32, i32p, f32p, f32p, C.c_float, C.c_int32, C.c_uint32, f32p]),
    "apt_medium_probe": (C.c_int, [C.c_int32, C.c_int32, i32p, f32p, C.c_int32, f32p, C.c_uint32, f32p]),
    "apt_emitter_probe": (C.c_int, [C.c_void_p, C.c_int32, f32p, C.c_uint32, f32p]),
    "apt_texture_probe": (C.c_int, [C.c_void_p, C.c_int32, i32p, f32p, f32p]),
    "apt_renderer_info": (C.c_int, [C.c_void_p, i32p, i32p, C.POINTER(C.c_int64), i32p, C.POINTER(C.c_char_p), i32p]),
    "apt_measure_sclk_mhz": (C.c_int, [C.c_int32, f32p]),
    "apt_last_error": (C.c_char_p, []),
    "apt_version": (C.c_char_p, []),
}

_libs = {}


def use(variant: str) -> str:
    """Make `variant` ("fast" | "exact") the library `load()` hands out from now on; returns the previous choice.  Objects created
    earlier keep the library they were created with."""
    global _variant, LIB_PATH
    if variant not in LIB_PATHS:
        raise ValueError(f"unknown build variant {variant!r}")
    prev, _variant = _variant, variant
    LIB_PATH = LIB_PATHS[variant]
    return prev


def variant() -> str:
    return _variant


def load(variant: str | None = None):
    """dlopen the in-tree library (of the current, or the given, variant) and bind every declared symbol (raises if any is missing)."""
    v = variant or _variant
    if v in _libs:
        return _libs[v]
    path = LIB_PATHS[v]
    if not os.path.exists(path):
        raise AptError(f"{path} is missing: build it with `python -m adapt_amd.build` (needs hipcc); "
                       "adapt_amd has no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    tag = lib.apt_version()
    overridden = os.environ.get("ADAPT_MI_LIB" if v == "fast" else "ADAPT_MI_LIB_EXACT")      # only the variant whose path was replaced skips the check
    if f"arithmetic: {v}".encode() not in tag and not overridden:
        raise AptError(f"{path} reports {tag!r}: not the {v} build")
    _libs[v] = lib
    return lib


def arithmetic(lib=None) -> str:
    """"fast" or "exact": what the loaded library says about itself (apt_version)."""
    tag = (lib or load()).apt_version().decode()
    return "exact" if "arithmetic: exact" in tag else "fast"


def check(rc: int, what: str = "", lib=None):
    if rc != 0:
        msg = (lib or load()).apt_last_error()
        raise AptError(f"{what or 'adapt_mi'} failed ({rc}): {msg.decode() if msg else '?'}")
