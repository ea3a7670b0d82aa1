"""Systematic difference between the fast and the exact build? high-spp renders of one scene: fast(seed 0), exact(seed 0), exact(seed 1);
block-averaged relative differences against the exact/seed-1 noise floor."""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from adapt_amd.parsers import scene_parsing
from adapt_amd.renderer import Renderer
sdir, xml, spp = sys.argv[1], sys.argv[2], int(sys.argv[3])
mb = int(sys.argv[4]) if len(sys.argv) > 4 else None
S = int(sys.argv[5]) if len(sys.argv) > 5 else None
tup = scene_parsing(sdir, xml)
w, h = 64, 48
def run(exact, seed):
    r = Renderer(*tup, width=w, height=h, exact=exact, seed=seed, max_bounce=mb, num_shadow_ray=S)
    r.render(n_spp=spp); img = r.pixels.to_numpy().astype(np.float64); st = r.stats(); r.close()
    return img, st
f0, sf = run(False, 0); e0, se = run(True, 0); e1, s1 = run(True, 1); f1, sf1 = run(False, 1)
def blocks(a): return a.reshape(w // 8, 8, h // 8, 8, 3).mean(axis=(1, 3))
def rel(a, b): return float(np.mean((a - b) ** 2 / (b ** 2 + 1e-2)))
print("n_shade fast/exact seed0", sf["n_shade"], se["n_shade"], "rel diff %.2e" % ((sf["n_shade"] - se["n_shade"]) / se["n_shade"]), "| seed1 fast/exact %.2e" % ((sf1["n_shade"] - s1["n_shade"]) / s1["n_shade"]), "| exact seed0 vs seed1 %.2e" % ((se["n_shade"] - s1["n_shade"]) / s1["n_shade"]))
print("mean radiance fast %.6f exact %.6f exact(seed1) %.6f fast(seed1) %.6f" % (f0.mean(), e0.mean(), e1.mean(), f1.mean()))
print("relMSE fast-vs-exact same seed %.3e | exact seed0-vs-seed1 (noise) %.3e | fast0-vs-exact1 %.3e | fast1-vs-exact0 %.3e" % (rel(f0, e0), rel(e0, e1), rel(f0, e1), rel(f1, e0)))
bf, be, b1 = blocks(f0), blocks(e0), blocks(e1)
print("8x8-block max rel diff: fast-vs-exact %.3e, exact-vs-exact(seed1) %.3e" % (np.abs(bf - be).max() / be.mean(), np.abs(be - b1).max() / be.mean()))
