"""Grid volumes for the volumetric tracer: the `<volume>` element of a scene file -> a density grid and the record the kernels read.

Host-side counterpart of the reference's `GridVolume_np` (bxdf/volume.py:36-218) and of its `.vol` loader
(bxdf/vol_loader/vol2numpy.cpp:35-73).  What the reference can actually render is one configuration: a one-channel Mitsuba `.vol`
file with `mono2rgb = true` (the grid is tripled and tinted with a colour ramp along z, the volume becomes RGB).  A plain `mono`
volume fails upstream when its majorant is exported (`vec3([maj, maj, maj])` with a 3-vector `maj`, volume.py:177) and a three-channel
file fails in the density scaling (volume.py:104,118); both are refused here with that explanation instead of being guessed at.
"""
from __future__ import annotations

import os
import struct
import xml.etree.ElementTree as xet
from typing import Tuple

import numpy as np

from .parsers.general_parser import get, rgb_parse, transform_parse

__all__ = ["read_vol", "GridVolume_np"]

_PHASE_IDS = {"hg": 0, "multi-hg": 1, "rayleigh": 2, "mie": 3, "transparent": -1}       # bxdf/medium.py:25


def read_vol(path: str) -> Tuple[np.ndarray, Tuple[int, int, int, int]]:
    """Mitsuba `.vol` (version 3, float32 encoding) -> (float32 [z][y][x][channel], (xres, yres, zres, channels))"""
    with open(path, "rb") as fh:
        raw = fh.read()
    if len(raw) < 48 or raw[:3] != b"VOL" or raw[3] != 3:
        raise ValueError(f"{path}: not a version-3 .vol file")
    encoding, xres, yres, zres, channels = struct.unpack_from("<5i", raw, 4)
    if encoding != 1:
        raise ValueError(f"{path}: only float32 voxels (encoding 1) are supported, got encoding {encoding}")
    if channels not in (1, 3):
        raise ValueError(f"{path}: {channels} channels; supported: 1, 3")
    count = xres * yres * zres * channels
    if xres <= 0 or yres <= 0 or zres <= 0 or len(raw) < 48 + 4 * count:
        raise ValueError(f"{path}: truncated or empty grid ({xres} x {yres} x {zres} x {channels})")
    grid = np.frombuffer(raw, "<f4", count=count, offset=48).astype(np.float32).reshape(zres, yres, xres, channels)
    return grid, (xres, yres, zres, channels)


def _z_colour_ramp(zres: int) -> np.ndarray:
    """the tint `mono2rgb` applies along z (volume.py:138-159): cyan-ish -> white over the first third, white -> yellow-ish after"""
    first = zres // 3
    fall = np.linspace(1, 0, first, dtype=np.float32) ** 0.65
    rise = np.linspace(0, 1, zres - first, dtype=np.float32) ** 0.6
    ramp = np.ones((zres, 3), np.float32)
    ramp[:first, 0] = 1 - fall
    ramp[first:, 2] = 1 - rise
    return ramp


class GridVolume_np:
    """`<volume type="mono" phase_type="hg" ...>`; attributes keep the reference's names."""

    def __init__(self, elem: xet.Element):
        kind = elem.get("type")
        if kind not in ("mono", "rgb", "none"):
            raise NotImplementedError(f"GridVolume type '{kind}' is not supported.")
        phase = elem.get("phase_type")
        if phase not in _PHASE_IDS:
            raise NotImplementedError(f"Phase function type '{phase}' is not supported.")
        self.type_name, self.phase_type, self.phase_type_id = kind, phase, _PHASE_IDS[phase]
        self.albedo = np.ones(3, np.float32)
        self.density_scaling = np.ones(3, np.float32)
        self.par = np.zeros(3, np.float32)
        self.pdf = np.float32([1., 0., 0.])
        self.mono2rgb = False
        self.rotation, self.offset, self.scale = np.eye(3, dtype=np.float32), np.zeros(3, np.float32), None
        path = None
        for node in elem:
            name = node.get("name")
            if node.tag == "rgb" and name in ("albedo", "density_scaling", "par", "pdf"):
                setattr(self, name, rgb_parse(node))
            elif node.tag == "bool" and name == "mono2rgb":
                self.mono2rgb = node.get("value") in {"True", "true"}
            elif node.tag == "string" and name == "density_grid":
                path = get(node, "path", str)
            elif node.tag == "transform" and name == "toWorld":
                self.rotation, self.offset, self.scale = transform_parse(node)
        if kind == "none":
            raise NotImplementedError("a <volume type='none'> declares nothing to render; remove it")
        if path is None or not os.path.exists(path):
            raise RuntimeError(f"Volume file not found: {path}")
        grid, (self.xres, self.yres, self.zres, channels) = read_vol(path)
        if channels != 1:
            raise NotImplementedError("three-channel .vol files fail upstream in the density scaling (bxdf/volume.py:104,118); use a one-channel file")
        if not self.mono2rgb:
            raise NotImplementedError("a mono volume without mono2rgb fails upstream when its majorant is exported (bxdf/volume.py:177); "
                                      "set <bool name=\"mono2rgb\" value=\"true\"/>")
        self.type_id, self.channel = 2, 3                                     # GridVolume_np.RGB
        grid = np.concatenate([grid, grid, grid], axis=-1) * _z_colour_ramp(self.zres)[:, None, None, :]
        self.density_grid = np.ascontiguousarray(grid * self.density_scaling, np.float32)
        if self.rotation is None:
            self.rotation = np.eye(3, dtype=np.float32)
        if self.offset is None:
            self.offset = np.zeros(3, np.float32)
        scale = np.eye(3, dtype=np.float32) if self.scale is None else np.diag(self.scale)
        self.forward_t = self.rotation @ scale                                # float64 when the rotation comes from scipy, as upstream

    def get_shape(self):
        return (self.zres, self.yres, self.xres)

    def get_aabb(self):
        x, y, z = self.xres, self.yres, self.zres
        corners = np.float32([[0, 0, 0], [x, 0, 0], [0, y, 0], [x, y, 0], [0, 0, z], [x, 0, z], [0, y, z], [x, y, z]])
        world = corners @ self.forward_t.T + self.offset
        return world.min(axis=0) - 0.01, world.max(axis=0) + 0.01

    def get_majorant(self, guard=0.2, scale_ratio=1.05):
        maj = self.density_grid.max(axis=(0, 1, 2))
        maj = np.maximum(maj, np.mean(maj) * guard)
        maj *= scale_ratio
        return np.float32(maj)

    def pack(self):
        """-> (int32[5] type, xres, yres, zres, phase type ; float32[33] albedo, inv_T (row-major), trans, mini, maxi, majorant,
        majorant pdf, phase par, phase lobe weights ; float32 grid [z][y][x][3])"""
        lo, hi = self.get_aabb()
        maj = self.get_majorant()
        floats = np.concatenate([self.albedo, np.float32(np.linalg.inv(self.forward_t)).reshape(-1), np.float32(self.offset), np.float32(lo),
                                 np.float32(hi), maj, np.float32(maj / maj.sum()), self.par, self.pdf]).astype(np.float32)
        ints = np.int32([self.type_id, self.xres, self.yres, self.zres, self.phase_type_id])
        return ints, floats, self.density_grid

    def __repr__(self):
        return f"<Volume grid RGB (mono2rgb) with phase {self.phase_type}, {self.xres} x {self.yres} x {self.zres}>"
