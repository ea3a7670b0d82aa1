"""Per-object descriptor handed to the renderer constructor.

Attribute names follow reference `parsers/obj_desc.py:28-65` (`tri_num`,
`meshes`, `normals`, `vns`, `uv_coords`, `bsdf`, `aabb`, `emitter_ref_id`,
`type`, `R`, `t`, `texture_group`) because `Renderer.__init__` consumers read
them by name.
"""
from __future__ import annotations

import numpy as np

__all__ = ["get_aabb", "ObjDescriptor"]


def get_aabb(meshes: np.ndarray, _type: int = 0) -> np.ndarray:
    """(2,3) float32 bounds.  Mesh: min/max over vertices, any axis thinner than
    1e-3 is padded by +-2e-2; sphere record: centre -+ radius (obj_desc.py:9-25)."""
    if _type != 0:
        return np.float32((meshes[0, 0] - meshes[0, 1], meshes[0, 0] + meshes[0, 1]))
    lo = meshes.min(axis=1).min(axis=0)
    hi = meshes.max(axis=1).max(axis=0)
    thin = ~(np.abs(hi - lo) > 1e-3)
    lo[thin] -= 2e-2
    hi[thin] += 2e-2
    return np.float32((lo, hi))


class ObjDescriptor:
    def __init__(self, meshes, normals, bsdf, vert_normal=None, uv_coords=None, texture_group=None,
                 R=None, t=None, emit_id=-1, _type=0):
        self.tri_num = meshes.shape[0]
        self.meshes = meshes
        self.uv_coords = uv_coords
        self.normals = normals
        self.vns = vert_normal
        self.R = R
        self.t = t
        self.bsdf = bsdf
        self.texture_group = texture_group
        self.aabb = get_aabb(meshes, _type)
        self.emitter_ref_id = emit_id
        self.type = _type

    def __repr__(self):
        kind = "sphere" if self.type == 1 else f"mesh[{self.tri_num}]"
        return f"<{kind} aabb={self.aabb.tolist()} emitter={self.emitter_ref_id}>"
