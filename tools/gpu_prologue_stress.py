"""Stress of the staged-ray prologue (shade_stage.hpp fix_prologue; ADVICE r5: its load order is outside the HSA memory model).
APT_FLAT_DEFER_ALL=1 sends EVERY ray of a render through the staging lists of all 32 sub-queues, so every launch of every bounce runs the
claim / serve / wait protocol on every sub-queue.  A lost hand-over is a lost path: the image of a full-size render must be bit-identical
from run to run (the pipeline is deterministic) and every path must be accounted for.  Prints one line per scene."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
os.environ["APT_FLAT_DEFER_ALL"] = "1"
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer

REPEATS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for sdir, xml, w, h, spp in (("cbox", "c2_cbox.xml", 512, 512, 64), ("cbox", "glass_box.xml", 256, 256, 64), ("csphere", "c3_balls_mono.xml", 256, 256, 32)):
    tup = scene_parsing(os.path.join(ROOT, "scenes", sdir), xml)
    imgs, stats = [], []
    for k in range(REPEATS):
        r = Renderer(*tup, width=w, height=h, num_shadow_ray=1)
        r.render(n_spp=spp)
        imgs.append(r.color.to_numpy()); st = r.stats(); stats.append({k_: st[k_] for k_ in ("n_samples", "n_extend", "n_shade", "n_shadow", "n_lit", "n_draws")})
        r.close()
    same_img = all(np.array_equal(imgs[0], im, equal_nan=True) for im in imgs[1:])
    same_st = all(stats[0] == s for s in stats[1:])
    print(f"{xml}: {w}x{h}x{spp} spp, every ray staged, {REPEATS} renders: images bit-identical {same_img}, statistics identical {same_st}, "
          f"n_samples {stats[0]['n_samples']} (= {w * h * spp}: {stats[0]['n_samples'] == w * h * spp}), n_shade {stats[0]['n_shade']}", flush=True)
    assert same_img and same_st and stats[0]["n_samples"] == w * h * spp
