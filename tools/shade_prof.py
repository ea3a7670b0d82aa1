"""k_shade section profile (GPU box; library built with -DAPT_SHADE_PROF: tools/build_variant.sh prof -DAPT_SHADE_PROF, ADAPT_MI_LIB=build_exp/libadapt_mi_prof.so)
    python tools/shade_prof.py <scene dir> <xml> <w> <h> <spp> <bounces> "<S values>" """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("APT_LANES", "1")
from adapt_amd.parsers import scene_parsing
from adapt_amd.renderer import Renderer
sdir, xml, w, h, spp, mb = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
tup = scene_parsing(sdir, xml)
for S in [int(x) for x in sys.argv[7].split()]:
    r = Renderer(*tup, width=w, height=h, max_bounce=mb, num_shadow_ray=S, profile=True)
    r.render(n_spp=spp); r.synchronize()
    print(f"--- S={S}", file=sys.stderr, flush=True)
    try:
        r.stats()
    except Exception as e:          # the profile slots overlay two statistics the call checks
        pass
    r.close()
