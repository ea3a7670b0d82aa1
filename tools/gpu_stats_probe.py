#!/usr/bin/env python3
"""HIP vs oracle path statistics for variants of a scene (isolates which rule makes the counts differ)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
from adapt_amd.scene_pack import make_config, pack_scene
from oracle import binding as ob

d, f = sys.argv[1], sys.argv[2]
for label, edit, kw in (("as is", {}, {}), ("no rr", {"use_rr": False}, {}), ("bounce1", {}, {"max_bounce": 1}), ("bounce2", {}, {"max_bounce": 2}), ("bounce3", {}, {"max_bounce": 3}),
                        ("S0", {}, {"num_shadow_ray": 0})):
    em, arr, objs, cfg = scene_parsing(d, f)
    cfg = dict(cfg); cfg.update(edit)
    rdr = Renderer(em, arr, objs, cfg, width=64, height=48, **kw)
    rdr.render(n_spp=4)
    st = rdr.stats()
    rc = make_config(cfg, width=64, height=48, **kw)
    _, _, ost = ob.OracleScene(pack_scene(em, arr, objs, cfg), rc.cam_t).render(rc, 4)
    print(f"{label:8s} hip shade {st['n_shade']} extend {st['n_extend']} draws {st['n_draws']} shadow {st['n_shadow']} | oracle shade {ost['n_shade']} draws {ost['n_draws']} shadow {ost['n_shadow']}")
    rdr.close()
