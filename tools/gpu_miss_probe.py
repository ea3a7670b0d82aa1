#!/usr/bin/env python3
"""For every continuation ray the oracle sees MISS, ask the HIP closest-hit kernel what it finds."""
import sys
import numpy as np
sys.path.insert(0, ".")
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
from adapt_amd.scene_pack import make_config, pack_scene
from oracle import binding as ob

d, f, w, h = sys.argv[1], sys.argv[2], 64, 48
parsed = scene_parsing(d, f)
rdr = Renderer(*parsed, width=w, height=h)
fs = pack_scene(*parsed); rc = make_config(parsed[3], width=w, height=h)
osc = ob.OracleScene(fs, rc.cam_t)
O, D, meta = [], [], []
for i in range(w):
    for j in range(h):
        col, ev, nd = osc.trace_sample(rc, i, j, 1)
        for b, e in enumerate(ev):
            O.append(e[12:15]); D.append(e[15:18]); meta.append((i, j, b, int(e[0]), e[9:12].copy()))
O, D = np.float32(O), np.float32(D)
obj, prim, t, uv, ns = osc.intersect(O, D)
hp, ht, huv = rdr.intersect(O, D)
bad = np.argwhere((prim != hp) | ((t != ht) & (prim >= 0))).reshape(-1)
print("continuation rays", len(O), "disagreements", len(bad))
for k in bad[:12]:
    print(meta[k], "o", O[k], "d", D[k], "|d|", np.linalg.norm(D[k]), "oracle prim/t", prim[k], t[k], "hip prim/t", hp[k], ht[k])
