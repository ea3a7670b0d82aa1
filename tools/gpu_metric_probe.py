#!/usr/bin/env python3
"""Image metrics HIP vs oracle (same stream) for one scene: tools/gpu_metric_probe.py <scene_dir> <xml> [spp]"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
from adapt_amd.scene_pack import make_config, pack_scene
from oracle import binding as ob
from conftest import image_metrics

d, f = sys.argv[1], sys.argv[2]; spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
w, h = 64, 48
parsed = scene_parsing(d, f)
r = Renderer(*parsed, width=w, height=h); r.render(n_spp=spp)
rc = make_config(parsed[3], width=w, height=h)
ref, cnt, ost = ob.OracleScene(pack_scene(*parsed), rc.cam_t).render(rc, spp)
acc = r.color.to_numpy()
m = image_metrics(acc / spp, ref / spp)
bad = np.argwhere(np.abs(acc / spp - ref / spp).max(axis=-1) > 1e-3 * (1 + np.abs(ref / spp).max(axis=-1)))
print(f, m, "stats", {k: (r.stats()[k], ost[k]) for k in ("n_shade", "n_shadow", "n_draws")}, "bad pixels", len(bad))
