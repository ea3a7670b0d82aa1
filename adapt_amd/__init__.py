"""adapt_amd — MI355X-native (gfx950) wavefront path tracer behind AdaPT's `pt` renderer contract.

    from adapt_amd import scene_parsing, Renderer
    rdr = Renderer(*scene_parsing("scenes/cbox", "c2_cbox.xml"))
    rdr.render(n_spp=64); img = rdr.pixels.to_numpy()

The render path is hand-written HIP reached through the C-ABI in include/adapt_mi.h
(adapt_amd/libadapt_mi.so, built by `python -m adapt_amd.build`); there is no CPU fallback.
"""
from .parsers.xml_parser import scene_parsing  # noqa: F401

__all__ = ["scene_parsing", "Renderer", "VolumeRenderer", "load_renderer"]
__version__ = "0.1.0"


def __getattr__(name):
    if name == "Renderer":
        from .renderer import Renderer
        return Renderer
    if name == "VolumeRenderer":
        from .renderer import VolumeRenderer
        return VolumeRenderer
    raise AttributeError(name)


def load_renderer(scene_dir: str, xml: str, **kw):
    """scene_parsing + Renderer in one call; keyword arguments go to Renderer."""
    from .renderer import Renderer
    return Renderer(*scene_parsing(scene_dir, xml), **kw)
