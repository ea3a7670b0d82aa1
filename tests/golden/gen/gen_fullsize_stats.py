#!/usr/bin/env python3
"""tests/golden/fullsize_{c2,c3}.npz: statistics of the ORACLE's render of BASELINE configs[1] / [2] at their full film size
(512 x 512; 64 spp of the 1024; all bounces) - means of the 8 x 8 grid of 64 x 64-pixel tiles, per-pixel image at 1/8 resolution
(box filter), and the path statistics.  Lets an un-gated GPU test check the FULL frame of C2 / C3 against the oracle on the same Philox
stream without minutes of host time on the GPU box (the oracle is pinned to the reference by the other fixtures).

    python tests/golden/gen/gen_fullsize_stats.py          (authoring container; ~3 min on 8 threads; no /root/reference needed)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, ROOT)
from adapt_amd.parsers import scene_parsing          # noqa: E402
from adapt_amd.scene_pack import make_config, pack_scene   # noqa: E402
from oracle import binding as ob                        # noqa: E402

CASES = {"c2": ("cbox", "c2_cbox.xml", 8), "c3": ("csphere", "c3_balls_mono.xml", 16)}
SPP, W, H = 64, 512, 512


def reduce(img, f):
    return img.reshape(W // f, f, H // f, f, 3).mean(axis=(1, 3))


if __name__ == "__main__":
    for tag, (sdir, xml, bounces) in CASES.items():
        tup = scene_parsing(os.path.join(ROOT, "scenes", sdir), xml)
        rc = make_config(tup[3], width=W, height=H, max_bounce=bounces)
        osc = ob.OracleScene(pack_scene(*tup), rc.cam_t)
        t = time.time()
        acc, cnt, st = osc.render(rc, SPP, threads=ob.num_threads())
        img = (acc / np.float32(cnt)).astype(np.float64)
        fin = np.isfinite(img).all(axis=2)
        img[~fin] = 0.0                                  # (C2's one inf pixel appears beyond 64 spp; kept out of the means anyway)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"fullsize_{tag}.npz"), tiles=np.float32(reduce(img, 64)), small=np.float32(reduce(img, 8)),
                            non_finite=np.int32((~fin).sum()), spp=np.int32(SPP), width=np.int32(W), height=np.int32(H), max_bounce=np.int32(bounces),
                            **{k: np.int64(st[k]) for k in ("n_samples", "n_shade", "n_shadow", "n_draws")})
        print(f"fullsize_{tag}: {W}x{H}x{SPP} spp, {bounces} bounces, {time.time() - t:.0f} s; mean {img.mean():.6f}, n_shade {st['n_shade']}, n_draws {st['n_draws']}")
