"""Wavefront OBJ reader + mesh transforms for the scene front end.

The reference delegates OBJ reading to the third-party `pywavefront>=1.3.0`
(`parsers/obj_loader.py:34-62`, absent from this image) and only slices the
interleaved per-face-vertex stream it returns.  This module reads the OBJ text
itself and produces the same four arrays:

    meshes  (N,3,3) float32   triangle vertices, face order, vertex order kept
    normals (N,3)   float32   geometric normal = normalize((v1-v0) x (v2-v1))
    vns     (N,3,3) float32   per-face-vertex normals, or None if the file has no `vn`
    uvs     (N,3,2) float32   per-face-vertex uvs,     or None if the file has no `vt`

Polygons are fan-triangulated (0, i, i+1), the order pywavefront emits.  Like the reference, which reads the vertex stream of the
FIRST material of the file only (`for wrapper in obj.materials.values(): ...; break`, obj_loader.py:35-37), faces of every other
material are dropped - with a warning here, silently upstream.  "First" is pywavefront's (1.3) order of `obj.materials`: the `newmtl`
entries of a `mtllib` in definition order, entered when the `mtllib` line is read; then materials first met in a `usemtl` line; faces
before any `usemtl` belong to a default material entered when the first such face is read.  So an OBJ whose .mtl defines [B, A] and
whose faces use A first loads B's faces, upstream and here.  (A `mtllib` file that cannot be read is skipped here; pywavefront raises.)
None of the bundled meshes has more than one material.  `apply_transform` /
`calculate_surface_area` follow reference obj_loader.py:82-122 including the
float32/float64 mixing of numpy (rotation matrices are float64, so a rotated
mesh is float64 until the final pack casts it back).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

__all__ = ["read_obj", "extract_obj_info", "apply_transform", "calculate_surface_area", "TRIANGLE_MESH", "SPHERE"]

TRIANGLE_MESH = 0
SPHERE = 1


def _resolve(idx: int, count: int) -> int:
    """OBJ indices are 1-based; negative ones count from the end."""
    return idx - 1 if idx > 0 else count + idx


def read_obj(path: str):
    """Parse `v`/`vt`/`vn`/`f` records.  Returns (positions, uvs|None, normals|None)
    as per-face-vertex float32 arrays of shape (N,3,k)."""
    import os
    pos, tex, nrm = [], [], []
    faces = {}                                   # material key -> (tri_p, tri_t, tri_n); key None = faces before any `usemtl`
    order = []                                   # pywavefront's `obj.materials` order (module docstring)
    group = None
    with open(path, "r") as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line[0] == "#":
                continue
            head, _, rest = line.partition(" ")
            if head == "mtllib":
                for name in rest.split():
                    try:
                        with open(os.path.join(os.path.dirname(path), name), "r") as mf:
                            for ml in mf:
                                mh, _, mr = ml.strip().partition(" ")
                                if mh == "newmtl" and mr.strip() not in order:
                                    order.append(mr.strip())
                    except OSError:
                        pass
                continue
            if head == "usemtl":
                group = rest.strip()
                if group not in order:
                    order.append(group)
                continue
            if head == "v":
                pos.append([float(x) for x in rest.split()[:3]])
            elif head == "vt":
                tex.append([float(x) for x in rest.split()[:2]])
            elif head == "vn":
                nrm.append([float(x) for x in rest.split()[:3]])
            elif head == "f":
                if group not in order:
                    order.append(group)          # (only the default material can get here)
                tri_p, tri_t, tri_n = faces.setdefault(group, ([], [], []))
                corners = []
                for tok in rest.split():
                    parts = tok.split("/")
                    vi = _resolve(int(parts[0]), len(pos))
                    ti = _resolve(int(parts[1]), len(tex)) if len(parts) > 1 and parts[1] else -1
                    ni = _resolve(int(parts[2]), len(nrm)) if len(parts) > 2 and parts[2] else -1
                    corners.append((vi, ti, ni))
                for k in range(1, len(corners) - 1):
                    tri = (corners[0], corners[k], corners[k + 1])
                    tri_p.append([pos[c[0]] for c in tri])
                    tri_t.append([tex[c[1]] if c[1] >= 0 else [0., 0.] for c in tri])
                    tri_n.append([nrm[c[2]] if c[2] >= 0 else [0., 0., 0.] for c in tri])
    if not faces:
        raise ValueError(f"OBJ file '{path}' contains no faces")
    first = order[0]
    if first not in faces:
        raise ValueError(f"OBJ file '{path}': its first material '{first}' owns no faces - the reference reads that material's (empty) "
                         "vertex stream (parsers/obj_loader.py:35-37); put the used material first in the .mtl")
    tri_p, tri_t, tri_n = faces[first]
    dropped = sum(len(v[0]) for k, v in faces.items() if k != first)
    if not tri_p:
        raise ValueError(f"OBJ file '{path}' contains no faces")
    if dropped:
        import warnings
        warnings.warn(f"{path}: {dropped} triangles of other materials ignored - the reference loads the first material's faces only "
                      "(parsers/obj_loader.py:35-37)", RuntimeWarning)
    meshes = np.float32(tri_p).reshape(-1, 3, 3)
    uvs = np.float32(tri_t).reshape(-1, 3, 2) if tex else None
    vns = np.float32(tri_n).reshape(-1, 3, 3) if nrm else None
    return meshes, uvs, vns


def extract_obj_info(path: str, verbose: bool = False):
    """Reference `extract_obj_info` contract (obj_loader.py:21-80):
    -> (meshes, geometric normals, vertex normals | None, uvs | None)."""
    meshes, uvs, vns = read_obj(path)
    e01 = meshes[:, 1, :] - meshes[:, 0, :]
    e12 = meshes[:, 2, :] - meshes[:, 1, :]
    normals = np.cross(e01, e12)
    normals /= np.linalg.norm(normals, axis=-1, keepdims=True)
    return meshes, normals, vns, uvs


def calculate_surface_area(meshes: np.ndarray, _type: int = TRIANGLE_MESH):
    """Sum of triangle areas, or 4 pi r^2 for a sphere record (obj_loader.py:82-93)."""
    if _type == SPHERE:
        radius = meshes[0, 1, 0]
        return 4. * np.pi * radius ** 2
    total = 0.
    for face in meshes:
        total += np.linalg.norm(np.cross(face[1] - face[0], face[2] - face[0])) / 2.
    return total


def apply_transform(meshes: np.ndarray, normals: Optional[np.ndarray], trans_r, trans_t, trans_s) -> Tuple[np.ndarray, np.ndarray]:
    """Rotate about the mesh centroid by RIGHT-multiplication, then translate
    (obj_loader.py:100-122).  Vertex normals are deliberately left alone, as in
    the reference.  Scale is parsed but never applied there either."""
    if trans_s is not None and not (trans_s[0] == trans_s[1] == trans_s[2]):
        trans_s[1] = trans_s[0]
        trans_s[2] = trans_s[0]
    if trans_r is not None:
        centroid = meshes.mean(axis=1).mean(axis=0)
        meshes -= centroid
        meshes = meshes @ trans_r
        if normals is not None:
            normals = normals @ trans_r
        meshes += centroid
    if trans_t is not None:
        meshes += trans_t
    return meshes, normals
