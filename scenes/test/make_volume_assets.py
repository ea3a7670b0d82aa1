"""Synthetic density grids for the grid-volume test scenes, written in the Mitsuba .vol layout the reference loads
(bxdf/vol_loader/vol2numpy.cpp:35-73: 'VOL' 3, int32 encoding = 1, xres, yres, zres, channels, 24 bytes of bounding box, float32 data
in [z][y][x][channel] order).  Deterministic: a few Gaussian puffs on a small grid.

    python scenes/test/make_volume_assets.py          # rewrites scenes/test/vol/*.vol
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def puffs(xres, yres, zres, centres):
    z, y, x = np.meshgrid(np.arange(zres), np.arange(yres), np.arange(xres), indexing="ij")
    d = np.zeros((zres, yres, xres), np.float64)
    for cx, cy, cz, r, a in centres:
        d += a * np.exp(-(((x - cx) / r) ** 2 + ((y - cy) / r) ** 2 + ((z - cz) / r) ** 2))
    d[d < 0.02] = 0.0                      # empty space around the puffs: null collisions only
    return np.float32(d)


def write_vol(path, grid):
    zres, yres, xres = grid.shape
    with open(path, "wb") as f:
        f.write(b"VOL\x03")
        f.write(struct.pack("<5i", 1, xres, yres, zres, 1))
        f.write(struct.pack("<6f", 0, 0, 0, 1, 1, 1))
        f.write(np.ascontiguousarray(grid, "<f4").tobytes())


if __name__ == "__main__":
    os.makedirs(os.path.join(HERE, "vol"), exist_ok=True)
    write_vol(os.path.join(HERE, "vol", "puffs_20x16x12.vol"), puffs(20, 16, 12, [(6, 8, 6, 3.5, 1.0), (13, 7, 5, 3.0, 0.8), (10, 11, 8, 2.5, 0.6)]))
    write_vol(os.path.join(HERE, "vol", "column_8x24x8.vol"), puffs(8, 24, 8, [(4, 5, 4, 2.5, 0.9), (4, 12, 4, 2.0, 1.0), (3.5, 19, 4.5, 2.2, 0.7)]))
    print("wrote", sorted(os.listdir(os.path.join(HERE, "vol"))))
