#!/usr/bin/env python3
"""C2 rate with the framebuffer read back to host memory after every step (the only host buffer the boundary returns)."""
import sys, time
sys.path.insert(0, ".")
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
r = Renderer(*scene_parsing("scenes/cbox", "c2_cbox.xml"), width=512, height=512, max_bounce=8)
r.render(n_spp=1024); r.pixels.to_numpy(); r.clear()
t = time.perf_counter()
for _ in range(3):
    r.render(n_spp=1024); img = r.pixels.to_numpy()
dt = time.perf_counter() - t
t = time.perf_counter(); img = r.pixels.to_numpy(); rb = time.perf_counter() - t
print(f"with readback every step: {512*512*1024*3/dt/1e6:.1f} Msamples/s; one readback (divide kernel + {img.nbytes/2**20:.1f} MiB D2H): {rb*1e3:.2f} ms")
