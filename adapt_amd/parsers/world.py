"""<world> element: background colours + free-space medium (reference
`parsers/world.py:21-47`).  The pt path reads only `medium.ior`
(path_tracer.py:456,478,493)."""
from __future__ import annotations

import xml.etree.ElementTree as xet

import numpy as np

from ..materials import Medium_np
from .general_parser import rgb_parse

__all__ = ["World_np"]


class World_np:
    def __init__(self, elem: xet.Element | None):
        self.skybox = np.zeros(3, np.float32)
        self.ambient = np.zeros(3, np.float32)
        medium_elem = None
        if elem is not None:
            for node in elem.findall("rgb"):
                if hasattr(self, node.get("name")):
                    setattr(self, node.get("name"), rgb_parse(node))
            medium_elem = elem.find("medium")
        self.medium = Medium_np(medium_elem, is_world=True)
        self.C = 1.0

    def export(self):
        """The reference exports a Taichi struct here; the HIP path only needs the ior."""
        return self

    def __repr__(self):
        return f"<World free space [{self.medium.type_name}] ior {self.medium.ior:.3f}>"
