"""Put the stand-ins + the read-only reference on sys.path and import the reference's pt modules.

Authoring-container only: /root/reference does not exist on the GPU box.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
REFERENCE = os.environ.get("ADAPT_REFERENCE", "/root/reference")


def setup():
    if not os.path.isdir(REFERENCE):
        raise RuntimeError(f"reference tree not found at {REFERENCE}")
    for p in (REFERENCE, os.path.join(HERE, "shim"), REPO):
        if p not in sys.path:
            sys.path.insert(0, p)
    warnings.filterwarnings("ignore")
    np.seterr(all="ignore")
    import rich.console
    rich.console.Console.log = lambda *a, **k: None         # the reference logs a lot
    rich.console.Console.print = lambda *a, **k: None
    import taichi as ti
    if os.environ.get("ADAPT_REF_MICROFACET") == "1":
        _enable_microfacet()
    return ti


class _FlagFlipLoader:
    """Imports bxdf/brdf.py with its module constant `__ENABLE_MICROFACET__` set to True: the switch the reference tells its users
    to flip by hand (brdf.py:8,60-63).  The source is read from the reference tree and compiled in memory; nothing is written."""
    def __init__(self, path):
        self.path = path

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        src = open(self.path).read()
        flag = "__ENABLE_MICROFACET__ = False"
        if src.count(flag) != 1:
            raise RuntimeError("bxdf/brdf.py: the microfacet switch is not where it used to be")
        module.__file__ = self.path
        exec(compile(src.replace(flag, "__ENABLE_MICROFACET__ = True"), self.path, "exec"), module.__dict__)


class _FlagFlipFinder:
    def find_spec(self, name, path=None, target=None):
        if name != "bxdf.brdf":
            return None
        import importlib.util
        return importlib.util.spec_from_loader(name, _FlagFlipLoader(os.path.join(REFERENCE, "bxdf", "brdf.py")))


def _enable_microfacet():
    if not any(isinstance(f, _FlagFlipFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _FlagFlipFinder())


def f32_constants(obj):
    """Taichi bakes Python-scope floats captured by a kernel as f32 constants; do the same for
    every float attribute of the data-oriented renderer object."""
    for k, v in list(vars(obj).items()):
        if isinstance(v, (float, np.floating)):
            setattr(obj, k, np.float32(v))


def make_renderer(scene_dir, xml, overrides=None, volumetric=False):
    """reference scene_parsing -> Renderer (or VolumeRenderer), with optional sensor overrides (width/height/max_bounce...)."""
    from parsers.xml_parser import scene_parsing
    if volumetric:
        from renderer.vpt import VolumeRenderer as Renderer
    else:
        from renderer.vanilla_renderer import Renderer
    base = scene_dir if os.path.isabs(scene_dir) else os.path.join(REFERENCE, "scenes", scene_dir)
    emitters, array_info, objs, cfg = scene_parsing(base, xml)
    for k, v in (overrides or {}).items():
        if k in ("width", "height"):
            cfg["film"][k] = v
        else:
            cfg[k] = v
    rdr = Renderer(emitters, array_info, objs, cfg)
    f32_constants(rdr)
    return rdr, (emitters, array_info, objs, cfg)
