"""GPU probe: volumetric tracer vs the CPU oracle on the reference-run vpt fixtures (same Philox stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import VPT_SCENE_TAGS, scene_from_golden
from adapt_amd.renderer import VolumeRenderer
from adapt_amd.scene_pack import make_config, pack_scene
from oracle import binding as ob

for tag in VPT_SCENE_TAGS:
    tup, g = scene_from_golden(tag, "vptscene")
    for (w, h, spp) in ((int(g["width"]), int(g["height"]), int(g["spp"])), (96, 72, 16)):
        rc = make_config(tup[3], width=w, height=h, volumetric=True)
        osc = ob.OracleScene(pack_scene(*tup), rc.cam_t)
        ref, cnt, ost = osc.render(rc, spp)
        r = VolumeRenderer(*tup, width=w, height=h)
        t = time.time(); r.render(n_spp=spp); acc = r.color.to_numpy(); dt = time.time() - t
        st = r.stats()
        a, b = acc.astype(np.float64) / spp, ref.astype(np.float64) / spp
        rel = float(np.mean((a - b) ** 2 / (b ** 2 + 1e-2)))
        within = float(np.mean(np.all(np.abs(a - b) <= 1e-3 * (1 + np.abs(b)), axis=2)))
        print(tag, (w, h, spp), r.info()["traversal"], "relMSE %.3g within %.4f max_abs %.3g" % (rel, within, float(np.abs(a - b).max())),
              {k: (st[k], ost.get(k)) for k in ("n_samples", "n_extend", "n_shade", "n_shadow", "n_lit", "n_draws", "n_track")}, "%.2fs" % dt, flush=True)
        if w == int(g["width"]):
            refrun = g["accum"].astype(np.float64) / spp
            print("   vs reference run: within %.4f" % float(np.mean(np.all(np.abs(a - refrun) <= 1e-3 * (1 + np.abs(refrun)), axis=2))))
        r.close()
