"""Host-side emitter records for the `pt` hot path.

Mirrors reference `emitters/abtract_source.py:246-281` (LightSource) and the
four exporters `emitters/{point,area,spot,collimated}.py`.  `pack()` yields the
flat record the device `sample_hit / eval_le / solid_angle_pdf` need
(abtract_source.py:44-54): type, bool_bits, obj_ref_id | intensity, dir, pos,
inv_area, r.
"""
from __future__ import annotations

import xml.etree.ElementTree as xet

import numpy as np

from .parsers.general_parser import get, rgb_parse, vec3d_parse

__all__ = ["LightSource", "PointSource", "AreaSource", "SpotSource", "CollimatedSource", "SOURCE_MAP",
           "POINT_SOURCE", "AREA_SOURCE", "SPOT_SOURCE", "COLLIMATED_SOURCE"]

POINT_SOURCE = 0
AREA_SOURCE = 1
SPOT_SOURCE = 2
COLLIMATED_SOURCE = 4
DEG2RAD = np.pi / 180.


class LightSource:
    type_id = -1

    def __init__(self, elem: xet.Element):
        self.intensity = np.ones(3, np.float32)
        for node in elem.findall("rgb"):
            name = node.get("name")
            if name == "emission":
                self.intensity = rgb_parse(node)
            elif name == "scaler":
                self.intensity *= rgb_parse(node)
        self.type = elem.get("type")
        self.id = elem.get("id")
        self.inv_area = 1.0
        self.attached = False
        self.in_free_space = True
        self.emit_time = 0.0
        flag = elem.find("boolean")
        if flag is not None and flag.get("value").lower() == "false":
            self.in_free_space = False
        self.pos = np.zeros(3, np.float32)
        self.dir = np.float32([0, 0, 1])
        self.r = 0.0

    # bool_bits: bit0 position-delta, bit1 direction-delta, bit2 area, bit4 in free space
    def bool_bits(self) -> int:
        raise NotImplementedError

    def pack(self):
        ints = np.int32([self.type_id, self.bool_bits(), -1, 0])
        flts = np.concatenate([self.intensity, self.dir, self.pos, [self.inv_area, self.r]]).astype(np.float32)
        return ints, flts

    def _read_pose(self, elem: xet.Element):
        points = elem.findall("point")
        assert len(points) >= 2
        for p in points:
            name = p.get("name")
            if name in ("position", "pos"):
                self.pos = vec3d_parse(p)
            elif name in ("direction", "dir"):
                self.dir = vec3d_parse(p)
                n = np.linalg.norm(self.dir)
                if n < 1e-5:
                    raise ValueError(f"Direction of source <{self.id}> is ill-conditioned.")
                self.dir /= n

    def __repr__(self):
        return f"<{self.type.capitalize()} light source. Intensity: {self.intensity}. Attached = {self.attached}>"


class PointSource(LightSource):
    type_id = POINT_SOURCE

    def __init__(self, elem: xet.Element):
        super().__init__(elem)
        centre = elem.find("point")
        assert centre is not None
        self.pos = vec3d_parse(centre)

    def bool_bits(self):
        return 0x01 + (int(self.in_free_space) << 4)


class AreaSource(LightSource):
    """Must be attached to a shape through <ref type="emitter">; inv_area is
    filled in by the scene parser (xml_parser.py:56-64)."""
    type_id = AREA_SOURCE

    def __init__(self, elem: xet.Element):
        super().__init__(elem)
        self.attached = True

    def bool_bits(self):
        return (int(self.in_free_space) << 4) | 0x04


class SpotSource(LightSource):
    type_id = SPOT_SOURCE

    def __init__(self, elem: xet.Element):
        super().__init__(elem)
        self._read_pose(elem)
        self.half_cos = np.cos(15.0 * DEG2RAD)
        for f in elem.findall("float"):
            if f.get("name") == "half-angle":
                self.half_cos = np.cos(max(1e-3, get(f, "value", float)) * DEG2RAD)
        self.r = self.half_cos          # device `r` = cos(half angle) for spots (spot.py:51)
        self.inv_area = 1.0

    def bool_bits(self):
        return 0x01 + (int(self.in_free_space) << 4)


class CollimatedSource(LightSource):
    type_id = COLLIMATED_SOURCE

    def __init__(self, elem: xet.Element):
        super().__init__(elem)
        self._read_pose(elem)
        self.radius = 0.
        for f in elem.findall("float"):
            if f.get("name") == "radius":
                self.radius = max(0., get(f, "value", float))
        self.r = self.radius
        self.inv_area = 1 if self.radius == 0 else (1. / np.pi) / (self.radius * self.radius)

    def bool_bits(self):
        return int(self.radius == 0) + 0x02 + (int(self.in_free_space) << 4)


SOURCE_MAP = {"point": PointSource, "area": AreaSource, "spot": SpotSource, "collimated": CollimatedSource}
