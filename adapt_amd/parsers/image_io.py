"""Minimal image readers for texture files: binary PPM (P6) and 8-bit non-interlaced PNG.

The reference reads textures with OpenCV (`cv.imread` + BGR->RGB, bxdf/texture.py:61-62); OpenCV is not a
dependency here.  `imread_rgb` returns what that pair returns: an (H, W, 3) uint8 RGB array.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

__all__ = ["imread_rgb", "write_ppm"]


def _read_ppm(data: bytes) -> np.ndarray:
    tokens, pos = [], 2
    while len(tokens) < 3:                      # width, height, maxval; '#' comments allowed between them
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tokens.append(int(data[pos:end])); pos = end
    w, h, maxval = tokens
    if maxval != 255:
        raise ValueError("PPM: only maxval 255 is supported")
    pos += 1                                    # the single whitespace byte after maxval
    return np.frombuffer(data, np.uint8, w * h * 3, pos).reshape(h, w, 3).copy()


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _read_png(data: bytes) -> np.ndarray:
    pos, chunks, ihdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            chunks.append(body)
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = ihdr
    channels = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if depth != 8 or interlace != 0 or channels is None:
        raise ValueError("PNG: only 8-bit, non-interlaced grey / RGB / RGBA files are supported")
    raw = zlib.decompress(b"".join(chunks))
    stride = w * channels
    img = np.zeros((h, stride), np.int32)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        f = raw[y * (stride + 1)]
        line = np.frombuffer(raw, np.uint8, stride, y * (stride + 1) + 1).astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - channels] if x >= channels else 0
                b = prev[x]
                c = prev[x - channels] if x >= channels else 0
                pred = a if f == 1 else ((a + b) >> 1 if f == 3 else _paeth(int(a), int(b), int(c)))
                cur[x] = (line[x] + pred) & 255
        img[y] = cur; prev = cur
    img = img.astype(np.uint8).reshape(h, w, channels)
    if channels == 1:
        return np.repeat(img, 3, axis=2)
    if channels == 2:
        return np.repeat(img[..., :1], 3, axis=2)
    return img[..., :3].copy()


def imread_rgb(path: str) -> np.ndarray:
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:2] == b"P6":
        return _read_ppm(data)
    if data[:8] == b"\x89PNG\r\n\x1a\n":
        return _read_png(data)
    raise ValueError(f"unsupported texture image format: {path} (binary PPM and 8-bit PNG are read)")


def write_ppm(path: str, rgb: np.ndarray) -> None:
    rgb = np.ascontiguousarray(rgb, np.uint8)
    with open(path, "wb") as fh:
        fh.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        fh.write(rgb.tobytes())
