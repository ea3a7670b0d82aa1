"""Stand-in for the reference's compiled `vol_loader` extension (bxdf/vol_loader/vol2numpy.cpp needs pybind11 headers the image's
reference tree does not carry): reads a Mitsuba .vol file with numpy, same return convention - (flat float32 array, (xres, yres, zres,
channels)).  Generator use only."""
import struct

import numpy as np


def vol_file_to_numpy(path, force_mono_color=False):
    with open(path, "rb") as f:
        raw = f.read()
    assert raw[:3] == b"VOL" and raw[3] == 3, "not a .vol v3 file"
    enc, xres, yres, zres, ch = struct.unpack_from("<5i", raw, 4)
    assert enc == 1 and ch in (1, 3)
    data = np.frombuffer(raw, "<f4", count=xres * yres * zres * ch, offset=4 + 20 + 24).copy()
    if force_mono_color and ch == 3:
        data = data.reshape(-1, 3)[:, 1].copy(); ch = 1
    return data, (xres, yres, zres, ch)
