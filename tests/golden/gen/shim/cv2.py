"""Stand-in for the absent `opencv-python`, enough for the reference's texture loader (bxdf/texture.py:61-67):
imread of the binary PPM / 8-bit PNG files this repo's test scenes use (BGR channel order, as OpenCV returns),
cvtColor(BGR2RGB).  `resize` is not provided: test textures stay below the loader's 2048-pixel limit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", "..")))
from adapt_amd.parsers.image_io import imread_rgb  # noqa: E402

COLOR_BGR2RGB = 4


def imread(path, *a, **k):
    if not os.path.exists(path):
        return None
    return np.ascontiguousarray(imread_rgb(path)[..., ::-1])


def cvtColor(img, code):
    assert code == COLOR_BGR2RGB
    return np.ascontiguousarray(img[..., ::-1])


def resize(*a, **k):
    raise NotImplementedError("cv2.resize stand-in: textures above 2048 px are not used by the test scenes")
