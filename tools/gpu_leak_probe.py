import os, sys
sys.path.insert(0, os.getcwd())
import torch
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer, VolumeRenderer
def free_gb():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2 ** 30
tup = scene_parsing("scenes/cbox", "c2_cbox.xml")
tv = scene_parsing("scenes/test", "media_a.xml")
base = free_gb()
for rnd in range(4):
    for k in range(25):
        r = Renderer(*tup, width=96, height=64); r.render(n_spp=2); r.pixels.to_numpy(); r.close()
        v = VolumeRenderer(*tv, width=96, height=64); v.render(n_spp=2); v.pixels.to_numpy(); v.close()
    print("after", (rnd + 1) * 50, "cycles: drift %.3f GiB" % (base - free_gb()), flush=True)
