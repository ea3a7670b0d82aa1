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
}
ALL_TAGS = list(SCENES)


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
            cache[tag] = scene_parsing(d, f)
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
