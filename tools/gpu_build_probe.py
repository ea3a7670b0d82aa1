"""Scene-load time of the three BVH builders (host binned SAH vs device LBVH / PLOC) on the bunny-field stand-in at three sizes.
    python tools/gpu_build_probe.py        (GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adapt_amd.renderer import DeviceScene
from adapt_amd.scene_pack import pack_scene
from adapt_amd.synth import bunny_field

for levels in (2, 3, 4):
    fs = pack_scene(*bunny_field(levels=levels))
    for builder in ("sah", "lbvh", "ploc", "sah", "lbvh", "ploc"):
        os.environ["APT_BVH_BUILDER"] = builder
        t = time.perf_counter(); sc = DeviceScene(fs, 0); dt = time.perf_counter() - t
        sc.close()
        print(f"{fs.n_prims:8d} primitives  {builder:4s}  apt_scene_create {dt * 1e3:8.1f} ms (build + 8-wide collapse + upload of every table)", flush=True)
