"""Image readers for texture files: binary PPM (P6) and 8-bit non-interlaced PNG with the readers below, JPEG (what every textured scene
of the reference names: scenes/cbox/bunny.xml:176-186, kitchen, bathroom) and anything else through Pillow when it is importable.

The reference reads textures with OpenCV (`cv.imread` + BGR->RGB, bxdf/texture.py:61-62) and shrinks anything above 2048 px with
`cv.resize` (texture.py:65-68); OpenCV (`opencv-python>=4.6.0`, requirements.txt:9) is not a dependency here.  `imread_rgb` returns what the
first pair returns - an (H, W, 3) uint8 RGB array; for JPEG both OpenCV and Pillow decode through libjpeg(-turbo) with its defaults (slow
integer IDCT, fancy upsampling) - and `resize_bilinear_u8` restates what `cv.resize(img, (w, h))` computes for 8-bit images: INTER_LINEAR
with pixel-centre alignment in OpenCV's published fixed-point form (11-bit coefficients, modules/imgproc/src/resize.cpp: HResizeLinear /
VResizeLinear<uchar, int, short>).  No OpenCV here to run against: the restatement is pinned by its own known answers only (tests/test_parser.py:
identity, constant images, a hand-computed 4 -> 2 case).
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

__all__ = ["imread_rgb", "write_ppm", "resize_bilinear_u8", "write_jpeg"]


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


def _pil():
    try:
        from PIL import Image
        return Image
    except ImportError:
        return None


def imread_rgb(path: str) -> np.ndarray:
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:2] == b"P6":
        return _read_ppm(data)
    if data[:8] == b"\x89PNG\r\n\x1a\n":
        try:
            return _read_png(data)
        except ValueError:
            if _pil() is None:
                raise
    Image = _pil()
    if Image is None:
        kind = "JPEG" if data[:2] == b"\xff\xd8" else "this"
        raise ValueError(f"unsupported texture image format: {path} (binary PPM and 8-bit PNG are read without Pillow; {kind} files need Pillow, which is not importable)")
    import io
    with Image.open(io.BytesIO(data)) as im:
        return np.asarray(im.convert("RGB"), np.uint8).copy()          # (cv.imread drops alpha and expands grey to three channels as well)


def resize_bilinear_u8(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """`cv.resize(img, (new_w, new_h))` for an (H, W, C) uint8 image: bilinear, sample positions (x + 0.5) * scale - 0.5, coefficients
    rounded to 11 bits, rows combined as ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2 >> 2 (OpenCV's 8-bit path)."""
    h, w = img.shape[:2]
    if (w, h) == (new_w, new_h):
        return img.copy()

    def taps(n_src, n_dst):
        scale = n_src / n_dst
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        f = (f - i0).astype(np.float32)
        lo, hi = i0 < 0, i0 >= n_src - 1
        f[lo | hi] = 0.0
        i0 = np.clip(i0, 0, n_src - 1)
        i1 = np.minimum(i0 + 1, n_src - 1)
        c1 = np.rint(f.astype(np.float64) * 2048.0).astype(np.int64)           # cvRound: to nearest, ties to even
        c0 = np.rint((1.0 - f.astype(np.float64)) * 2048.0).astype(np.int64)
        return i0, i1, c0, c1
    x0, x1, a0, a1 = taps(w, new_w)
    y0, y1, b0, b1 = taps(h, new_h)
    src = img.astype(np.int64)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]      # horizontal pass: 8-bit x 11-bit coefficients
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def write_jpeg(path: str, rgb: np.ndarray, quality: int = 95) -> None:
    """--img_ext jpg (parsers/opts.py:25; the reference writes through ti.tools.imwrite).  Needs Pillow."""
    Image = _pil()
    if Image is None:
        raise RuntimeError("writing JPEG needs Pillow, which is not importable; use --img_ext png | bmp | npy")
    Image.fromarray(np.ascontiguousarray(rgb, np.uint8), "RGB").save(path, format="JPEG", quality=quality)


def write_ppm(path: str, rgb: np.ndarray) -> None:
    rgb = np.ascontiguousarray(rgb, np.uint8)
    with open(path, "wb") as fh:
        fh.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        fh.write(rgb.tobytes())
