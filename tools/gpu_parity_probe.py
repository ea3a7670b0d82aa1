#!/usr/bin/env python3
"""Quick on-GPU probe: HIP path vs CPU oracle on the bench scenes + a throughput number.
(Development aid; the judged checks are tests/ -m gpu and bench.py.)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
from adapt_amd.scene_pack import make_config, pack_scene
from oracle import binding as ob


def probe(d, f, w, h, spp, **kw):
    parsed = scene_parsing(d, f)
    rdr = Renderer(*parsed, width=w, height=h, profile=True, **kw)
    rdr.render(n_spp=spp)
    acc = rdr.color.to_numpy()
    st = rdr.stats()
    fs = pack_scene(*parsed)
    rc = make_config(parsed[3], width=w, height=h, **{k: v for k, v in kw.items() if k in ("max_bounce", "num_shadow_ray")})
    osc = ob.OracleScene(fs, rc.cam_t)
    ref, cnt, ost = osc.render(rc, spp, threads=0)
    diff = np.abs(acc - ref)
    rel = diff / (1 + np.abs(ref))
    out = {"scene": f, "w": w, "h": h, "spp": spp, "max_abs": float(diff.max()), "n_pix_diff": int((diff.max(axis=2) > 0).sum()),
           "n_pix": w * h, "frac_within_1e-3": float((rel.max(axis=2) <= 1e-3).mean()), "mean_hip": float(acc.mean()), "mean_ref": float(ref.mean()),
           "relMSE": float(np.mean((acc / spp - ref / spp) ** 2 / ((ref / spp) ** 2 + 1e-2))),
           "stats_hip": {k: st[k] for k in ("n_samples", "n_shade", "n_shadow", "n_lit", "n_draws", "n_extend", "n_shadow_traced")},
           "stats_ref": ost, "kernel_ms": st["kernel_ms"], "render_ms": st["render_ms"]}
    print(json.dumps(out))
    rdr.close()
    return out


if __name__ == "__main__":
    probe("scenes/cbox", "c2_cbox.xml", 64, 64, 8)
    probe("scenes/cbox", "c2_cbox.xml", 256, 256, 16, max_bounce=4)
    probe("scenes/csphere", "c3_balls_mono.xml", 64, 64, 4)
    probe("scenes/cbox", "glass_box.xml", 64, 64, 4)
    # throughput: C2 shape, modest spp
    parsed = scene_parsing("scenes/cbox", "c2_cbox.xml")
    for prof in (False, True):
        rdr = Renderer(*parsed, profile=prof)
        rdr.render(n_spp=16); rdr.synchronize()
        t = time.time(); rdr.render(n_spp=128); rdr.synchronize(); dt = time.time() - t
        st = rdr.stats()
        print(json.dumps({"c2_msamples_per_s": 512 * 512 * 128 / dt / 1e6, "profile": prof, "kernel_ms": st["kernel_ms"], "render_ms": st["render_ms"],
                          "launches": st["launches"]}))
        rdr.close()
