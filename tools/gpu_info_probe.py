import sys; sys.path.insert(0, ".")
from bench import load_scene, CONFIGS
from adapt_amd.renderer import Renderer
for c in ("c4", "c5"):
    sdir, sfile, W, H, spp, bounces, label = CONFIGS[c]
    r = Renderer(*load_scene(sdir, sfile), width=W, height=H, max_bounce=bounces)
    print(c, r.info()); r.close()
