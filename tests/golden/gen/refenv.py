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
    return ti


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
