#!/usr/bin/env python3
"""Find pixels whose finiteness differs between HIP and the oracle and print the oracle's trace of the offending sample."""
import sys
import numpy as np
sys.path.insert(0, ".")
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
from adapt_amd.scene_pack import make_config, pack_scene
from oracle import binding as ob

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
parsed = scene_parsing("scenes/cbox", "c2_cbox.xml")
rdr = Renderer(*parsed)
rdr.render(n_spp=spp)
a = rdr.color.to_numpy()
rc = make_config(parsed[3])
osc = ob.OracleScene(pack_scene(*parsed), rc.cam_t)
b, _, _ = osc.render(rc, spp)
bad = np.argwhere(np.isfinite(a).all(axis=2) != np.isfinite(b).all(axis=2))
print("mismatching pixels:", bad.tolist(), "hip nonfinite:", int((~np.isfinite(a)).any(axis=2).sum()), "oracle nonfinite:", int((~np.isfinite(b)).any(axis=2).sum()))
for i, j in bad[:3]:
    print("pixel", i, j, "hip", a[i, j], "oracle", b[i, j])
    for cnt in range(1, spp + 1):
        col, ev, nd = osc.trace_sample(rc, int(i), int(j), cnt)
        if not np.isfinite(col).all() or not np.isfinite(ev).all():
            print("  sample", cnt, "colour", col, "draws", nd)
            for e in ev:
                print("     obj", int(e[0]), "prim", int(e[1]), "t", e[2], "direct", e[3:6], "emit", e[6:9], "thr", e[9:12])
# also compare per-sample for that pixel through 1-spp HIP renders would need a re-render; the trace is usually enough
