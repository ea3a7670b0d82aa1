#!/usr/bin/env python3
"""Closest-hit / occlusion parity on a full-size synthetic scene (HIP BVH mode vs the oracle's restated reference BVH)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from adapt_amd.renderer import Renderer
from adapt_amd.scene_pack import make_config, pack_scene
from adapt_amd.synth import SYNTH_SCENES
from oracle import binding as ob

name = sys.argv[1] if len(sys.argv) > 1 else "three-bunnies"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
parsed = SYNTH_SCENES[name]()
r = Renderer(*parsed)
print(r.info())
rc = make_config(parsed[3])
sc = ob.OracleScene(pack_scene(*parsed), rc.cam_t, build_bvh=True)
rs = np.random.RandomState(11)
o = rs.uniform([0.2, 0.1, 0.2], [5.3, 5.3, 5.3], size=(n, 3)).astype(np.float32)
d = rs.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
prim, t, uv = r.intersect(o, d)
obj_o, prim_o, t_o, uv_o, _ = sc.intersect(o, d, use_bvh=True)
diff = prim != prim_o
print("rays", n, "prim mismatches", int(diff.sum()), "of which t equal", int((t[diff] == t_o[diff]).sum()), "t mismatches among same prim", int((t[~diff] != t_o[~diff]).sum()))
bad = np.argwhere(diff & (t != t_o)).reshape(-1)[:8]
for k in bad:
    print("  ray", k, "hip prim/t", prim[k], t[k], "oracle prim/t", prim_o[k], t_o[k], "o", o[k], "d", d[k])
tmax = rs.uniform(0.2, 6.0, n).astype(np.float32)
occ, occ_o = r.occluded(o, d, tmax), sc.occluded(o, d, tmax, use_bvh=True)
print("occlusion mismatches", int((occ != occ_o).sum()), "hip occluded", int(occ.sum()), "oracle occluded", int(occ_o.sum()))
