#!/usr/bin/env python3
"""Fast build (flat sweep) vs exact build vs oracle: hits on random rays, occlusion flags, small images, per scene."""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adapt_amd.parsers import scene_parsing
from adapt_amd.renderer import Renderer
from adapt_amd.scene_pack import make_config, pack_scene
from oracle import binding as ob

SC = {"cbox": ("scenes/cbox", "c2_cbox.xml"), "balls": ("scenes/csphere", "c3_balls_mono.xml"), "glass": ("scenes/cbox", "glass_box.xml"),
      "fa": ("scenes/test", "features_a.xml"), "fb": ("scenes/test", "features_b.xml"), "fc": ("scenes/test", "features_c.xml")}
for tag in (sys.argv[1:] or list(SC)):
    parsed = scene_parsing(*SC[tag])
    rf, re_ = Renderer(*parsed, width=96, height=96, exact=False), Renderer(*parsed, width=96, height=96, exact=True)
    print(tag, rf.info()["traversal"], rf.info()["arithmetic"], "|", re_.info()["traversal"], re_.info()["arithmetic"])
    rs = np.random.RandomState(7)
    n = 200000
    o = rs.uniform([0.1, 0.1, 0.1], [5.4, 5.3, 5.4], size=(n, 3)).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:200, 0] = 0.0; d[:200] /= np.linalg.norm(d[:200], axis=1, keepdims=True)
    tmax = rs.uniform(0.2, 8.0, n).astype(np.float32)
    pf, tf, uf = rf.intersect(o, d); pe, te, ue = re_.intersect(o, d)
    same = pf == pe
    hit = same & (pe >= 0)
    rel = np.abs(tf[hit] - te[hit]) / np.maximum(np.abs(te[hit]), 1e-3)
    print("  prim mismatches %d / %d (%.2e), of those |dt|/t max %.2e ; same-prim rel t err max %.2e mean %.2e ; uv abs err max %.2e" % (
        int((~same).sum()), n, (~same).mean(), (np.abs(tf[~same] - te[~same]) / np.maximum(np.abs(te[~same]), 1e-3)).max() if (~same).any() else 0.0,
        rel.max(), rel.mean(), np.abs(uf[hit] - ue[hit]).max()))
    bad = np.argwhere((~same) & (np.abs(tf - te) / np.maximum(np.abs(te), 1e-3) > 1e-4)).reshape(-1)[:6]
    for k in bad:
        print("    ray", k, "fast prim/t", pf[k], tf[k], "exact prim/t", pe[k], te[k], "o", o[k].tolist(), "d", d[k].tolist())
    of, oe = rf.occluded(o, d, tmax), re_.occluded(o, d, tmax)
    print("  occlusion mismatches %d / %d" % (int((of != oe).sum()), n))
    rc = make_config(parsed[3], width=96, height=96)
    osc = ob.OracleScene(pack_scene(*parsed), rc.cam_t)
    ref, cnt, ost = osc.render(rc, 64, threads=ob.num_threads())
    b = (ref / np.float32(cnt)).astype(np.float64)
    for name, r in (("fast", rf), ("exact", re_)):
        r.render(n_spp=64)
        a = r.pixels.to_numpy().astype(np.float64)
        fin = np.isfinite(a).all(axis=2) & np.isfinite(b).all(axis=2)
        af, bf = a[fin], b[fin]
        st = r.stats()
        print("  %-5s vs oracle 96x96x64spp: relMSE %.3g within1e-3 %.5f max %.3g | n_shade %d/%d n_shadow %d/%d n_draws %d/%d" % (
            name, np.mean((af - bf) ** 2 / (bf ** 2 + 1e-2)), np.mean(np.all(np.abs(af - bf) <= 1e-3 * (1 + np.abs(bf)), axis=1)), np.abs(af - bf).max(),
            st["n_shade"], ost["n_shade"], st["n_shadow"], ost["n_shadow"], st["n_draws"], ost["n_draws"]))
    rf.close(); re_.close()
