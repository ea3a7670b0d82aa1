"""Texture host objects and the atlas packer (reference: bxdf/texture.py:33-94, parsers/texture_packing.py:31-69).

One atlas per map kind ("albedo", "normal", "bump", "roughness"); every texture keeps its size and its offset inside
the atlas.  The reference packs with the third-party `rectpack`; lookups are relative to a texture's own offset and
never leave its rectangle (`query` wraps at `w - 1` / `h - 1`), so the layout is free — a shelf packer is used here.
Checkerboard textures are declared upstream but their lookup is not implemented there (texture.py:96 TODO): refused.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as xet
from typing import Dict, List, Optional, Tuple

import numpy as np

from .parsers.general_parser import get
from .parsers.image_io import imread_rgb, resize_bilinear_u8

__all__ = ["Texture_np", "image_packer", "parse_texture", "TEX_TAGS", "TEX_INVALID"]

TEX_TAGS = ("albedo", "normal", "bump", "roughness")
TEX_INVALID = -255          # Texture.type of an object without that map (texture.py:93)


class Texture_np:
    MODE_IMAGE = 0
    MODE_CHECKER = 1

    def __init__(self, elem: xet.Element, max_size: int = 2048, base_dir: Optional[str] = None):
        self.tag = elem.get("tag", "albedo")
        self.id = elem.get("id")
        self.type = elem.get("type")
        self.scale_u = 1.
        self.scale_v = 1.
        self.off_x = self.off_y = 0
        if self.type == "checkerboard":
            raise NotImplementedError("checkerboard textures have no lookup upstream (bxdf/texture.py:96); only image textures are supported")
        path = elem.find("string").get("value")
        if not os.path.exists(path) and base_dir is not None and os.path.exists(os.path.join(base_dir, path)):
            path = os.path.join(base_dir, path)                 # extension: also relative to the scene file
        if not os.path.exists(path):
            raise ValueError(f"Texture image input path '{path}' does not exist.")
        self.texture_path = path
        img = imread_rgb(path)
        self.h, self.w = int(img.shape[0]), int(img.shape[1])
        if self.h > max_size or self.w > max_size:              # texture.py:65-68: each side clamped on its own (aspect not kept), cv.resize defaults
            self.w, self.h = min(self.w, max_size), min(self.h, max_size)
            img = resize_bilinear_u8(img, self.w, self.h)
        self.texture_img = img.astype(np.float32) / 255.
        if self.tag == "bump":                                   # up axis z -> y (texture.py:69-71)
            self.texture_img[..., [1, 2]] = self.texture_img[..., [2, 1]]
        for node in elem.findall("float"):
            if node.get("name") in ("scale_u", "scale_v"):
                setattr(self, node.get("name"), get(node, "value"))

    def record(self) -> Tuple[np.ndarray, np.ndarray]:
        """(int32[5] type, off_x, off_y, w, h ; float32[2] scale_u, scale_v) — the fields of the reference's Texture struct"""
        return np.int32([Texture_np.MODE_IMAGE, self.off_x, self.off_y, self.w, self.h]), np.float32([self.scale_u, self.scale_v])


def image_packer(textures: List[Texture_np]):
    """-> (square float32 atlas, {id: texture}) with off_x / off_y filled in"""
    order = sorted(range(len(textures)), key=lambda k: (-textures[k].h, -textures[k].w, k))
    size = 64
    while True:
        x = y = shelf = 0
        ok = True
        for k in order:
            t = textures[k]
            if t.w > size:
                ok = False; break
            if x + t.w > size:
                x, y, shelf = 0, y + shelf, 0
            if y + t.h > size:
                ok = False; break
            t.off_x, t.off_y = x, y
            x += t.w; shelf = max(shelf, t.h)
        if ok:
            break
        size *= 2
        if size > 8192:
            raise ValueError("Texture image packing failed")
    atlas = np.zeros((size, size, 3), np.float32)
    table = {}
    for t in textures:
        atlas[t.off_y:t.off_y + t.h, t.off_x:t.off_x + t.w] = t.texture_img
        table[t.id] = t
    return atlas, table


def parse_texture(nodes: List[xet.Element], base_dir: Optional[str] = None):
    """<texture> nodes -> ({tag: atlas | None}, {tag: {id: Texture_np} | None}) or (None, None)   (xml_parser.py:196-216)"""
    if len(nodes) == 0:
        return None, None
    by_tag: Dict[str, List[Texture_np]] = {t: [] for t in TEX_TAGS}
    for node in nodes:
        by_tag[node.get("tag", "albedo")].append(Texture_np(node, base_dir=base_dir))
    imgs, infos = {}, {}
    for tag, lst in by_tag.items():
        imgs[tag], infos[tag] = image_packer(lst) if lst else (None, None)
    return imgs, infos
