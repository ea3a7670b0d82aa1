"""Where apt_scene_create spends its time, per builder (GPU box): APT_SCENE_TIMING=1 python tools/scene_timing.py [levels]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["APT_SCENE_TIMING"] = "1"
from adapt_amd.renderer import DeviceScene
from adapt_amd.scene_pack import pack_scene
from adapt_amd.synth import bunny_field
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fs = pack_scene(*bunny_field(levels=levels))
for builder in ("sah", "ploc", "ploc"):
    os.environ["APT_BVH_BUILDER"] = builder
    print(f"--- {fs.n_prims} primitives, {builder}", file=sys.stderr, flush=True)
    t = time.perf_counter(); sc = DeviceScene(fs, 0); dt = time.perf_counter() - t
    sc.close()
    print(f"    apt_scene_create total {dt * 1e3:.1f} ms", file=sys.stderr, flush=True)
