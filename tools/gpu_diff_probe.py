#!/usr/bin/env python3
"""Localise HIP-vs-oracle differences: 1 spp, list the worst pixels with the oracle's per-bounce trace."""
import sys
import numpy as np
sys.path.insert(0, ".")
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
from adapt_amd.scene_pack import make_config, pack_scene
from oracle import binding as ob

d, f, w, h = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
if d == "synth":
    from adapt_amd.synth import SYNTH_SCENES
    parsed = SYNTH_SCENES[f]()
else:
    parsed = scene_parsing(d, f)
rdr = Renderer(*parsed, width=w, height=h)
rdr.render(n_spp=1)
acc = rdr.color.to_numpy()
fs = pack_scene(*parsed); rc = make_config(parsed[3], width=w, height=h)
osc = ob.OracleScene(fs, rc.cam_t, build_bvh=rc.use_bvh)
ref, _, _ = osc.render(rc, 1)
diff = np.abs(acc - ref).max(axis=2)
order = np.argsort(-diff.reshape(-1))[:12]
print("pixels with |diff| > 1e-4:", int((diff > 1e-4).sum()), "of", w * h, "sum hip", float(acc.sum()), "sum ref", float(ref.sum()))
types = fs.bxdf_i[:, 0]; isb = fs.bxdf_i[:, 2]
for p in order:
    i, j = divmod(int(p), h)
    if diff[i, j] <= 1e-5: break
    col, ev, nd = osc.trace_sample(rc, i, j, 1)
    objs = [int(e[0]) for e in ev]
    print(f"pixel ({i},{j}) hip={acc[i,j]} ref={ref[i,j]} draws={nd}")
    for e in ev:
        o = int(e[0])
        print(f"    obj {o} type {'bsdf' if isb[o] else 'brdf'}{types[o]} prim {int(e[1])} t={e[2]:.5f} direct={e[3:6]} emit*w={e[6:9]} thr={e[9:12]}")
