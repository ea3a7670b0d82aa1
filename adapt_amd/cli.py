"""`render.py`-compatible command line for the `pt` and `vpt` renderers on MI355X.

Mirrors the reference driver's flags and control flow for `--type pt|vpt --no_gui`
(`render.py:65-166`, `parsers/opts.py:15-44`): scene parsing, optional checkpoint load,
`iter_num + 1` samples (the reference's `--no_gui` loop runs `range(iter_num + 1)`,
render.py:80-81,118), periodic checkpoint saves, summary, optional quantile normalisation
(`utils/watermark.py:28-29`), image file `<output_path><img_name>-<scene file stem>-pt.<ext>`.
A cropped render writes the crop window `[start_x:end_x, start_y:end_y]` of the (w, h, 3) image; the reference slices
`[start_y:end_y, start_x:end_x]` on that same (w, h, 3) array (`utils/watermark.py:25`), which is the intended window only for
square crops - a deliberate deviation.
Differences: there is no GUI and no Taichi (`--arch` is accepted and ignored unless it is not
one of the known names), only `--type pt` and `--type vpt` exist (homogeneous media), the "RENDERED WITH AdaPT" watermark is not
stamped (`--no_watermark` is accepted), PNG/BMP are written by a small built-in encoder.
"""
from __future__ import annotations

import argparse
import os
import pickle
import struct
import sys
import time
import zlib

import numpy as np

__all__ = ["get_options", "main", "write_image", "to_display"]


def get_options(argv=None):
    p = argparse.ArgumentParser(description="AdaPT-compatible path tracing driver (HIP / MI355X)", fromfile_prefix_chars="@")
    p.add_argument("--config", default=None, help="file with one `key = value` (or `--key value`) per line")
    p.add_argument("--iter_num", default=-1, type=int, help="Number of iterations (-1: the scene's iter_num or 2000)")
    p.add_argument("--normalize", default=0., type=float, help="Normalize the output picture with its <x> quantile value")
    p.add_argument("--output_freq", default=0, type=int)
    p.add_argument("--input_path", default="./scenes/", type=str)
    p.add_argument("--output_path", default="./outputs/", type=str)
    p.add_argument("--chkpt_path", default="./checkpoint/", type=str)
    p.add_argument("--img_name", default="pbr", type=str)
    p.add_argument("--img_ext", default="png", choices=["png", "jpg", "bmp", "npy"])      # parsers/opts.py:25 (jpg through Pillow)
    p.add_argument("--scene", default="cbox", type=str)
    p.add_argument("--name", default="c2_cbox.xml", type=str)
    p.add_argument("--arch", default="hip", choices=["hip", "cpu", "gpu", "vulkan", "cuda"], help="kept for CLI compatibility; rendering always uses HIP")
    p.add_argument("--save_iter", default=-1, type=int)
    p.add_argument("--type", default="pt", choices=["pt", "vpt", "bdpt", "ao"])
    p.add_argument("-p", "--profile", default=False, action="store_true")
    p.add_argument("--no_gui", default=False, action="store_true")
    p.add_argument("-d", "--debug", default=False, action="store_true")
    p.add_argument("-a", "--analyze", default=False, action="store_true")
    p.add_argument("-l", "--load", default=False, action="store_true")
    p.add_argument("--no_cache", default=False, action="store_true")
    p.add_argument("--no_save_fig", default=False, action="store_true")
    p.add_argument("--no_watermark", default=False, action="store_true")
    # extensions
    p.add_argument("--device", default=0, type=int)
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--width", default=None, type=int)
    p.add_argument("--height", default=None, type=int)
    p.add_argument("--max_bounce", default=None, type=int)
    argv = list(sys.argv[1:] if argv is None else argv)
    pre, _ = p.parse_known_args(argv)
    if pre.config:
        extra = []
        for line in open(pre.config):
            line = line.split("#")[0].strip()
            if not line:
                continue
            key, _, val = line.partition("=")
            key, val = key.strip().lstrip("-"), val.strip()
            if val.lower() in ("false", "0", "no", "off") and any(a.dest == key and a.nargs == 0 for a in p._actions):
                continue                    # a switch set to false: leave it off (configargparse accepts `no_gui = false`)
            extra += [f"--{key}"] + ([val] if val and val.lower() not in ("true",) else [])
        argv = extra + argv           # command line wins over the file
    return p.parse_args(argv)


def to_display(img: np.ndarray) -> np.ndarray:
    """(w, h, 3) [x][y], y up  ->  (rows, cols, 3) uint8, row 0 on top (what ti.tools.imwrite stores)."""
    a = np.clip(np.nan_to_num(img, nan=0.0, posinf=1.0, neginf=0.0), 0.0, 1.0)
    a = np.ascontiguousarray(np.swapaxes(a, 0, 1)[::-1])
    return (a * 255.0 + 0.5).astype(np.uint8)


def write_image(img: np.ndarray, path: str):
    ext = path.rsplit(".", 1)[-1].lower()
    if ext == "npy":
        np.save(path, img)
        return
    px = to_display(img)
    h, w, _ = px.shape
    if ext in ("jpg", "jpeg"):
        from .parsers.image_io import write_jpeg
        write_jpeg(path, px)
        return
    if ext == "png":
        def chunk(tag, data):
            return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
        raw = b"".join(b"\x00" + px[r].tobytes() for r in range(h))
        blob = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    else:   # bmp, bottom-up BGR rows padded to 4 bytes
        row = (3 * w + 3) & ~3
        body = b"".join(px[r, :, ::-1].tobytes() + b"\x00" * (row - 3 * w) for r in range(h - 1, -1, -1))
        blob = b"BM" + struct.pack("<IHHI", 54 + len(body), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, len(body), 2835, 2835, 0, 0) + body
    with open(path, "wb") as f:
        f.write(blob)


def _folder(path: str) -> str:
    os.makedirs(path, exist_ok=True)
    return path


def main(argv=None) -> int:
    opts = get_options(argv)
    if opts.type not in ("pt", "vpt"):
        print(f"--type {opts.type}: only the `pt` and `vpt` renderers exist in this build (bdpt/ao are outside its scope)", file=sys.stderr)
        return 2
    from .parsers.xml_parser import scene_parsing
    from .renderer import Renderer, VolumeRenderer
    if opts.type == "vpt":                      # render.py:33 rdr_mapping: "vpt" -> VolumeRenderer
        Renderer = VolumeRenderer
    t0 = time.time()
    emitters, array_info, objs, cfg = scene_parsing(os.path.join(opts.input_path, opts.scene), opts.name)
    print(f"[adapt_amd] scene '{opts.scene}/{opts.name}': {array_info['primitives'].shape[0]} primitives, {len(objs)} objects, "
          f"{len(emitters)} emitters, parsed in {time.time() - t0:.3f} s")
    rdr = Renderer(emitters, array_info, objs, cfg, device=opts.device, seed=opts.seed, profile=opts.profile,
                   width=opts.width, height=opts.height, max_bounce=opts.max_bounce)
    stem = opts.name[:-4]
    max_iter = (opts.iter_num if opts.iter_num > 0 else cfg.get("iter_num", 2000)) + 1          # render.py:80-81
    chk_file = os.path.join(_folder(opts.chkpt_path), f"{opts.img_name}-{stem}-{opts.type}.pkl")
    if opts.load:
        with open(chk_file, "rb") as f:
            rdr.load_check_point(pickle.load(f))
        print(f"[adapt_amd] recovered from check-point, elapsed counter: {rdr.cnt[None]}")
    print(f"[adapt_amd] path tracing with {rdr.max_bounce} bounce(s), {rdr.w}x{rdr.h}, {max_iter} samples, {rdr.info()}")

    def save_chk():
        with open(chk_file, "wb") as f:
            pickle.dump(rdr.get_check_point(), f, protocol=pickle.HIGHEST_PROTOCOL)

    done = 0
    t1 = time.time()
    try:
        while done < max_iter:
            step = max_iter - done
            if opts.save_iter > 0:
                # the reference saves before iteration i whenever i % save_iter == 0 (render.py:119-121)
                if done % opts.save_iter == 0:
                    save_chk()
                step = min(step, opts.save_iter - done % opts.save_iter)
            rdr.render(n_spp=step)
            done += step
            if opts.output_freq > 0 and done % opts.output_freq == 0 and done < max_iter:
                out_dir = _folder(os.path.join(opts.output_path, f"{opts.img_name}-{stem}-{opts.type}"))
                write_image(rdr.pixels.to_numpy(), os.path.join(out_dir, f"img_{done:05d}.{opts.img_ext}"))
        rdr.synchronize()
    except KeyboardInterrupt:
        if opts.save_iter > 0:
            save_chk()
        print("[adapt_amd] quit on keyboard interruption")
    dt = time.time() - t1
    rdr.summary()
    n = rdr.w * rdr.h * done
    print(f"[adapt_amd] {n / max(dt, 1e-9) / 1e6:.1f} Msamples/s ({done} spp in {dt:.3f} s)")
    if opts.profile:
        st = rdr.stats()
        for k, ms in st["kernel_ms"].items():
            print(f"[adapt_amd]   {k:9s} {st['launches'][k]:6d} launches {ms:10.3f} ms")
    img = rdr.pixels.to_numpy()
    if rdr.do_crop:
        img = img[rdr.start_x:rdr.end_x, rdr.start_y:rdr.end_y, :]
    print(f"[adapt_amd] pixel max value = {np.nanmax(img):.3f}")
    if opts.normalize > 0.9:
        img = img / np.quantile(img, opts.normalize)
    if not opts.no_save_fig:
        out = os.path.join(_folder(opts.output_path), f"{opts.img_name}-{stem}-{opts.type}.{opts.img_ext}")
        write_image(img, out)
        print(f"[adapt_amd] wrote {out}")
    rdr.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
