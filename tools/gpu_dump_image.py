#!/usr/bin/env python3
"""Render a scene at 64x48 and save the accumulated image: tools/gpu_dump_image.py <scene_dir> <xml> <spp> <out.npy>"""
import sys
import numpy as np
sys.path.insert(0, ".")
from adapt_amd import scene_parsing
from adapt_amd.renderer import Renderer
d, f, spp, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
r = Renderer(*scene_parsing(d, f), width=64, height=48); r.render(n_spp=spp)
np.save(out, r.color.to_numpy())
print(out, {k: v for k, v in r.stats().items() if isinstance(v, int)})
