"""Stand-in for the absent `pywavefront>=1.3.0`, just enough for reference parsers/obj_loader.py:34-62:
Wavefront(path, collect_faces=True).materials.values() -> objects with `.vertex_format` and a flat,
interleaved per-face-vertex `.vertices` list ([T2F_][N3F_]V3F).  OBJ text is read by the repo's own reader."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", "..")))
from adapt_amd.parsers.obj_loader import read_obj
import numpy as np

class _Material:
    def __init__(self, fmt, verts): self.vertex_format, self.vertices = fmt, verts

class Wavefront:
    def __init__(self, path, collect_faces=False, **kw):
        meshes, uvs, vns = read_obj(path)
        parts, cols = [], []
        if uvs is not None: parts.append("T2F"); cols.append(uvs.reshape(-1, 2))
        if vns is not None: parts.append("N3F"); cols.append(vns.reshape(-1, 3))
        parts.append("V3F"); cols.append(meshes.reshape(-1, 3))
        self.materials = {"default": _Material("_".join(parts), np.concatenate(cols, axis=1).reshape(-1).tolist())}
