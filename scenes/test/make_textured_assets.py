#!/usr/bin/env python3
"""Deterministic assets of scenes/test/textured.xml: four small textures (binary PPM) and three meshes with `vt` records.
Run from the repository root; the outputs are committed."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from adapt_amd.parsers.image_io import write_ppm  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def textures():
    y, x = np.mgrid[0:32, 0:48].astype(np.float64)
    wood = np.stack([150 + 60 * np.sin(x * 0.7 + 3 * np.sin(y * 0.2)), 95 + 40 * np.sin(x * 0.7 + 3 * np.sin(y * 0.2) + 0.4), 50 + 25 * np.cos(y * 0.3)], -1)
    write_ppm(os.path.join(HERE, "tex", "wood.ppm"), np.clip(wood, 0, 255).astype(np.uint8))            # 48 x 32, neither square nor a power of two
    y, x = np.mgrid[0:16, 0:16]
    chk = np.where(((x // 4 + y // 4) % 2)[..., None] == 0, np.uint8([230, 230, 60]), np.uint8([40, 60, 200]))
    write_ppm(os.path.join(HERE, "tex", "tiles.ppm"), chk.astype(np.uint8))                             # 16 x 16
    y, x = np.mgrid[0:32, 0:32].astype(np.float64)
    nx, nz = 0.25 * np.sin(x * 0.6), 0.25 * np.cos(y * 0.45)
    ny = np.sqrt(np.clip(1 - nx * nx - nz * nz, 0, 1))
    nrm = np.stack([nx * 0.5 + 0.5, ny, nz * 0.5 + 0.5], -1) * 255                                       # read by the renderer as plain [0,1] components
    write_ppm(os.path.join(HERE, "tex", "ripples_normal.ppm"), np.clip(nrm, 0, 255).astype(np.uint8))   # 32 x 32
    y, x = np.mgrid[0:24, 0:24].astype(np.float64)
    bx, by = 0.2 * np.sin(x * 0.9) * np.cos(y * 0.5), 0.2 * np.sin(y * 0.8)
    bz = np.sqrt(np.clip(1 - bx * bx - by * by, 0, 1))
    bump = np.stack([bx * 0.5 + 0.5, by * 0.5 + 0.5, bz], -1) * 255                                      # z-up on disk; the loader swaps y and z for bump maps
    write_ppm(os.path.join(HERE, "tex", "dents_bump.ppm"), np.clip(bump, 0, 255).astype(np.uint8))      # 24 x 24


def obj(path, verts, uvs, normals, faces):
    with open(path, "w") as fh:
        for v in verts: fh.write("v %.6f %.6f %.6f\n" % tuple(v))
        for t in uvs: fh.write("vt %.6f %.6f\n" % tuple(t))
        for n in normals: fh.write("vn %.4f %.4f %.4f\n" % tuple(n))
        for f in faces: fh.write("f " + " ".join("%d/%d/%d" % c for c in f) + "\n")


def meshes():
    # back wall of the room (z = 5.592), normal -z, uv over the whole quad but not axis-symmetric
    obj(os.path.join(HERE, "meshes", "tex_back.obj"),
        [(5.496, 0, 5.592), (0, 0, 5.592), (0, 5.488, 5.592), (5.56, 5.488, 5.592)],
        [(0.05, 0.0), (1.0, 0.1), (0.9, 1.0), (0.0, 0.85)], [(0, 0, -1)],
        [((1, 1, 1), (2, 2, 1), (3, 3, 1)), ((1, 1, 1), (3, 3, 1), (4, 4, 1))])
    # a box standing on the floor, 12 triangles, every face with its own uv square
    x0, x1, y0, y1, z0, z1 = 0.9, 2.5, 0.0, 1.6, 2.6, 4.2
    c = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
    quads = [((0, 3, 2, 1), (0, 0, -1)), ((4, 5, 6, 7), (0, 0, 1)), ((0, 4, 7, 3), (-1, 0, 0)), ((1, 2, 6, 5), (1, 0, 0)), ((3, 7, 6, 2), (0, 1, 0)), ((0, 1, 5, 4), (0, -1, 0))]
    faces = []
    for k, (q, _n) in enumerate(quads):
        a, b, cc, d = [i + 1 for i in q]
        faces += [((a, 1, k + 1), (b, 2, k + 1), (cc, 3, k + 1)), ((a, 1, k + 1), (cc, 3, k + 1), (d, 4, k + 1))]
    obj(os.path.join(HERE, "meshes", "tex_box.obj"), c, [(0, 0), (1.5, 0), (1.5, 1.25), (0, 1.25)], [n for _, n in quads], faces)
    # a tilted panel (frosted), uv shifted outside [0,1] so that the wrap-around of the lookup is exercised, also negative
    obj(os.path.join(HERE, "meshes", "tex_panel.obj"),
        [(3.2, 0.3, 2.0), (4.8, 0.3, 2.6), (4.8, 2.4, 2.9), (3.2, 2.4, 2.3)],
        [(-0.4, -0.2), (1.7, -0.2), (1.7, 2.3), (-0.4, 2.3)], [(0.35, -0.13, -0.93)],
        [((1, 1, 1), (2, 2, 1), (3, 3, 1)), ((1, 1, 1), (3, 3, 1), (4, 4, 1))])


if __name__ == "__main__":
    textures(); meshes()
    print("assets written under", HERE)
