#!/usr/bin/env python3
"""Per-sample HIP-vs-oracle comparison: render one spp at a time, find (pixel, sample) pairs that differ, print the oracle trace."""
import sys
import numpy as np
sys.path.insert(0, ".")
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
from adapt_amd.scene_pack import make_config, pack_scene
from oracle import binding as ob

d, f, w, h, spp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
parsed = scene_parsing(d, f)
rdr = Renderer(*parsed, width=w, height=h)
fs = pack_scene(*parsed); rc = make_config(parsed[3], width=w, height=h)
osc = ob.OracleScene(fs, rc.cam_t)
types, isb = fs.bxdf_i[:, 0], fs.bxdf_i[:, 2]
prev = np.zeros((w, h, 3), np.float32)
shown = 0
hip_shade_prev = 0
for c in range(1, spp + 1):
    rdr.render(n_spp=1)
    acc = rdr.color.to_numpy()
    hip = acc - prev
    prev = acc
    ref, _, ost = osc.render(rc, 1, cnt=c - 1)
    st = rdr.stats()
    diff = np.abs(hip - ref).max(axis=2)
    print(f"cnt {c}: pixels differing > 1e-4: {(diff > 1e-4).sum()}   n_shade hip {st['n_shade'] - hip_shade_prev} oracle {ost['n_shade']}  draws oracle {ost['n_draws']}")
    hip_shade_prev = st["n_shade"]
    for p in np.argsort(-diff.reshape(-1))[:2]:
        i, j = divmod(int(p), h)
        if diff[i, j] <= 1e-4 or shown >= 4:
            break
        shown += 1
        col, ev, nd = osc.trace_sample(rc, i, j, c)
        print(f"  pixel ({i},{j}) hip={hip[i, j]} ref={ref[i, j]} draws={nd}")
        for e in ev:
            o = int(e[0])
            print(f"      obj {o} {'bsdf' if isb[o] else 'brdf'}{types[o]} prim {int(e[1])} t={e[2]:.6f} direct={e[3:6]} emit*w={e[6:9]} thr={e[9:12]}")
