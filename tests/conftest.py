"""Shared fixtures.  `-m "not gpu"` runs on the CPU-only authoring box (oracle vs golden vectors, host
logic, C-ABI surface); `-m gpu` runs on an MI355X and is where the HIP path is compared with the oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = {
    # tag: (directory, file, golden tag)   -- same inputs as the reference's cbox.xml / balls-mono.xml / complex.xml
    "cbox": (os.path.join(ROOT, "scenes", "cbox"), "c2_cbox.xml", "cbox"),
    "balls_mono": (os.path.join(ROOT, "scenes", "csphere"), "c3_balls_mono.xml", "balls_mono"),
    "glass_box": (os.path.join(ROOT, "scenes", "cbox"), "glass_box.xml", "complex"),
    # feature coverage (scenes/test): all five emitter types, the remaining surface models, the sensor flags
    "features_a": (os.path.join(ROOT, "scenes", "test"), "features_a.xml", "features_a"),
    "features_b": (os.path.join(ROOT, "scenes", "test"), "features_b.xml", "features_b"),
    "features_c": (os.path.join(ROOT, "scenes", "test"), "features_c.xml", "features_c"),
    # image textures (albedo / normal / bump maps on meshes); texture paths in the file are relative to the repository root
    "textured": (os.path.join(ROOT, "scenes", "test"), "textured.xml", "textured"),
    # Trowbridge-Reitz microfacet BRDFs: parsed with the reference's `__ENABLE_MICROFACET__` switch on (MICROFACET_TAGS below)
    "microfacet": (os.path.join(ROOT, "scenes", "test"), "microfacet.xml", "microfacet"),
}
ALL_TAGS = list(SCENES)
MICROFACET_TAGS = {"microfacet"}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with `-m gpu`")


def has_gpu() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def parsed():
    from adapt_amd.parsers import scene_parsing
    cache = {}

    def get(tag):
        if tag not in cache:
            d, f, _ = SCENES[tag]
            from adapt_amd import materials
            cwd, switch = os.getcwd(), materials.ENABLE_MICROFACET
            os.chdir(ROOT)                      # texture paths are relative to the repository root
            materials.ENABLE_MICROFACET = tag in MICROFACET_TAGS
            try:
                cache[tag] = scene_parsing(d, f)
            finally:
                os.chdir(cwd)
                materials.ENABLE_MICROFACET = switch
        return cache[tag]
    return get


@pytest.fixture(scope="session")
def flat(parsed):
    from adapt_amd.scene_pack import pack_scene
    cache = {}

    def get(tag):
        if tag not in cache:
            cache[tag] = pack_scene(*parsed(tag))
        return cache[tag]
    return get


@pytest.fixture(scope="session")
def oracle_scene(flat, parsed):
    from adapt_amd.scene_pack import make_config
    from oracle import binding as ob
    cache = {}

    def get(tag, build_bvh=False):
        key = (tag, build_bvh)
        if key not in cache:
            rc = make_config(parsed(tag)[3])
            cache[key] = ob.OracleScene(flat(tag), rc.cam_t, build_bvh=build_bvh)
        return cache[key]
    return get


def image_metrics(a, b):
    """a, b: (w,h,3) linear images.  relMSE as SURVEY §8(d) defines it, max-abs, fraction within 1e-3(1+|x|)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    rel = np.mean((a - b) ** 2 / (b ** 2 + 1e-2))
    within = np.mean(np.all(np.abs(a - b) <= 1e-3 * (1 + np.abs(b)), axis=2))
    return {"relMSE": float(rel), "max_abs": float(np.abs(a - b).max()), "frac_within": float(within)}


def record_metric(name, values):
    """Measured parity figures of a GPU test, one JSON line each, into the file APT_TEST_METRICS_LOG names (unset: nothing is written).
    The tolerances in the tests are these measurements with a stated margin; the log of record is profiles/r0N_parity_metrics.log."""
    path = os.environ.get("APT_TEST_METRICS_LOG")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in values.items()}}) + "\n")


# ---- the reference's own bundled scenes, as arrays (tests/golden/refscene_*.npz; the XML files stay in /root/reference)
VPT_SCENE_TAGS = sorted(f[len("vptscene_"):-4] for f in os.listdir(GOLDEN) if f.startswith("vptscene_") and f.endswith(".npz"))
REF_SCENE_TAGS = sorted(f[len("refscene_"):-4] for f in os.listdir(GOLDEN) if f.startswith("refscene_") and f.endswith(".npz"))


class _Packed:
    """host object that hands back its packed record (what adapt_amd.scene_pack asks of materials and emitters)"""
    def __init__(self, ints, floats):
        self._i, self._f = np.int32(ints), np.float32(floats)

    def pack(self):
        return self._i, self._f


class _Obj:
    def __init__(self, tri_num, kind, aabb, emitter_ref_id, bsdf):
        self.tri_num, self.type, self.aabb, self.emitter_ref_id, self.bsdf = int(tri_num), int(kind), np.float32(aabb), int(emitter_ref_id), bsdf


class _World:
    class medium:
        ior = 1.0


def scene_from_golden(tag, prefix="refscene"):
    """refscene_<tag>.npz / vptscene_<tag>.npz -> the 4-tuple `scene_parsing` returns (emitters, array_info, objects, prop) + the fixture"""
    g = golden(f"{prefix}_{tag}.npz")
    emitters = [_Packed(g["src_i"][k], g["src_f"][k]) for k in range(g["src_i"].shape[0])]
    objs = [_Obj(g["obj_info"][k, 1], g["obj_info"][k, 2], g["obj_aabb"][k], g["emitter_id"][k], _Packed(g["bxdf_i"][k], g["bxdf_f"][k]))
            for k in range(g["obj_info"].shape[0])]
    n = g["prims"].shape[0]
    arr = {"primitives": g["prims"], "n_g": g["normals"], "n_s": g["v_normals"], "uvs": np.zeros((n, 3, 2), np.float32), "indices": None}
    world = _World(); world.medium = type("M", (), {"ior": float(g["world_ior"])})()
    if "med_i" in g:                            # participating media, already flat
        n_obj = g["obj_info"].shape[0]
        world.medium.packed_medium = (int(g["med_i"][n_obj]), np.float32(g["med_f"][n_obj]))
        for k, o in enumerate(objs):
            if int(g["bxdf_i"][k, 2]):
                o.bsdf.medium = type("M", (), {"ior": float(g["med_f"][k, 0]), "packed_medium": (int(g["med_i"][k]), np.float32(g["med_f"][k]))})()
    prop = {"film": {"width": int(g["width"]), "height": int(g["height"])}, "fov": float(g["fov"]), "max_bounce": int(g["max_bounce"]),
            "num_shadow_ray": int(g["num_shadow_ray"]), "use_rr": bool(g["use_rr"]), "use_mis": bool(g["use_mis"]), "anti_alias": bool(g["anti_alias"]),
            "stratified_sampling": bool(g["stratified_sampling"]), "brdf_two_sides": bool(g["brdf_two_sides"]),
            "accelerator": "bvh" if int(g["accelerator_bvh"]) else "none", "rr_bounce_th": int(g["rr_bounce_th"]), "rr_threshold": float(g["rr_threshold"]),
            "transform": (np.float32(g["cam_dir"]), np.float32(g["cam_pos"]), None), "world": world, "has_vertex_normal": bool(g["has_vertex_normal"]),
            "packed_textures": None, "volume": []}
    return (emitters, arr, objs, prop), g
