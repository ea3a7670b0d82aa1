#!/usr/bin/env python3
"""Compare the accumulated image for different APT_LANES values (development aid)."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
parsed = scene_parsing("scenes/cbox", "c2_cbox.xml")
ref = None
for lanes in (1, 2, 3):
    for trial in range(2):
        os.environ["APT_LANES"] = str(lanes)
        r = Renderer(*parsed, width=96, height=64, spp_per_batch=2)
        r.render(n_spp=7); r.render(n_spp=6)
        img = r.color.to_numpy().copy(); st = {k: v for k, v in r.stats().items() if k.startswith("n_")}
        r.close()
        if ref is None: ref, rst = img, st
        d = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1)
        print("lanes", lanes, "trial", trial, "pixels differing from lanes=1:", int(d.sum()), "max abs", float(np.abs(img - ref).max()), "stats equal", st == rst)
