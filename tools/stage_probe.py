"""Stage-time probe (GPU box): one render lane, per-launch HIP events; prints the summed kernel times of a config for a few light-sample counts.
    APT_LANES=1 python tools/stage_probe.py <scene dir> <xml> <w> <h> <spp> <bounces> "<S values>" """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("APT_LANES", "1")
from adapt_amd.parsers import scene_parsing
from adapt_amd.renderer import Renderer
sdir, xml, w, h, spp, mb = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
tup = scene_parsing(sdir, xml)
for S in [int(x) for x in sys.argv[7].split()]:
    r = Renderer(*tup, width=w, height=h, max_bounce=mb, num_shadow_ray=S, profile=True)
    r.render(n_spp=4); r.synchronize(); r.clear()
    st0 = r.stats()
    r.render(n_spp=spp); r.synchronize()
    st = r.stats()
    k = {n: round(st["kernel_ms"][n] - st0["kernel_ms"][n], 2) for n in st["kernel_ms"]}
    print(f"S={S}: kernel_ms {k}  n_shade {st['n_shade'] - st0['n_shade']} n_shadow {st['n_shadow'] - st0['n_shadow']} draws {st['n_draws'] - st0['n_draws']}", flush=True)
    r.close()
