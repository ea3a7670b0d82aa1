"""Weak-scaling balance on a single GPU: every rank of an N-GPU step rendered one after the other (its interleaved bands, N x the
samples, as bench.py does per rank).  The step time of the real run is the slowest rank's: prints per-rank times and max / mean."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
tup = scene_parsing(os.path.join(ROOT, "scenes", "cbox"), "c2_cbox.xml")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for bw in (32, 16, 8, 4):
    times = []
    for rank in range(world):
        r = Renderer(*tup, width=512, height=512, max_bounce=8, rank=rank, world_size=world, band_width=bw)
        spp = 128 * world
        r.render(n_spp=spp); r.synchronize(); r.clear()
        t = time.perf_counter(); r.render(n_spp=spp); r.synchronize(); times.append(time.perf_counter() - t)
        r.close()
    mean = sum(times) / len(times)
    print("world", world, "band", bw, "per-rank ms", [round(1e3 * x, 1) for x in times], "max/mean %.3f" % (max(times) / mean), flush=True)
