"""Leaf-level XML value readers for the AdaPT scene format (version 1.1).

Host-side mirror of the reference's `parsers/general_parser.py:13-105`; same
function names and return conventions so scene files written for AdaPT load
unchanged.  dtype flow (float32 everywhere, scipy rotations in float64) is kept
on purpose: the packed geometry has to be bit-identical to what the reference
hands to its kernels.
"""
from __future__ import annotations

import xml.etree.ElementTree as xet
from typing import Optional, Tuple

import numpy as np
from scipy.spatial.transform import Rotation

__all__ = ["get", "parse_str", "rgb_parse", "vec3d_parse", "transform_parse", "parse_sphere_element"]


def get(node: xet.Element, name: str, _type=float):
    """Attribute `name` converted with `_type`; a missing attribute reads as "0"
    (reference general_parser.py:13-15)."""
    return _type(node.get(name, "0"))


def parse_str(val_str: str, no_else_branch: bool = False) -> np.ndarray:
    """"a, b, c" / "a b c" -> float32[3]; a lone scalar is broadcast to 3 unless
    `no_else_branch` (reference general_parser.py:17-27: comma is tried first)."""
    for sep in (",", " "):
        if sep in val_str:
            return np.float32([float(tok.strip()) for tok in val_str.split(sep)])
    if no_else_branch:
        raise ValueError("Value can not be a single digit, should be a vector splitted by ',' or [space]")
    return np.float32([float(val_str.strip())] * 3)


def rgb_parse(elem: Optional[xet.Element]) -> np.ndarray:
    """<rgb value="#RRGGBB" | "r,g,b" | "s"/> or <rgb r= g= b=/> (missing -> 0)
    (reference general_parser.py:29-46)."""
    if elem is None:
        raise ValueError("EmptyElementError: Element <RGB> is None.")
    text = elem.get("value")
    if text is None:
        if elem.get("r"):
            return np.float32([get(elem, "r"), get(elem, "g"), get(elem, "b")])
        raise ValueError("RGBError: RGB element does not contain valid field.")
    if text.startswith("#"):
        out = np.zeros(3, dtype=np.float32)
        for c in range(3):
            out[c] = int(text[1 + 2 * c:3 + 2 * c], 16) / 255.
        return out
    return parse_str(text)


def vec3d_parse(elem: xet.Element):
    """<point x= y= z=/> -> float32[3] (reference general_parser.py:48-54; the
    `value=` child form is kept for symmetry)."""
    if elem.tag == "point":
        if elem.find("value") is None:
            return np.float32([get(elem, "x"), get(elem, "y"), get(elem, "z")])
        return parse_str(elem.get("value"), no_else_branch=True)
    return None


def transform_parse(transform_elem: xet.Element) -> Tuple[Optional[np.ndarray], Optional[np.ndarray], Optional[np.ndarray]]:
    """<transform> children -> (rotation | look direction, translation | origin, scale).

    Reference general_parser.py:56-98.  Rotations are float64 scipy matrices
    (euler order "zxy", degrees); `lookat` returns the *direction vector* in the
    rotation slot and the origin in the translation slot (`up` is ignored)."""
    rot = trans = scale = None
    for child in transform_elem:
        tag = child.tag
        if tag == "translate":
            trans = np.float32([get(child, "x"), get(child, "y"), get(child, "z")])
        elif tag == "rotate":
            kind = child.get("type", "euler")
            if kind == "euler":
                angles = (get(child, "r"), get(child, "p"), get(child, "y"))
                rot = Rotation.from_euler("zxy", angles, degrees=True).as_matrix()
            elif kind == "quaternion":
                rot = Rotation.from_quat([get(child, "x"), get(child, "y"), get(child, "z"), get(child, "w")]).as_matrix()
            elif kind == "angle-axis":
                axis = np.float32([get(child, "x"), get(child, "y"), get(child, "z")])
                # NB: the reference divides (not multiplies) by the angle term
                # (general_parser.py:77); kept so existing scenes load identically.
                axis /= np.linalg.norm(axis) * get(child, "angle") / 180. * np.pi
                rot = Rotation.from_rotvec(axis).as_matrix()
            else:
                raise ValueError(f"Unsupported rotation representation '{kind}'")
        elif tag == "scale":
            scale = np.float32([get(child, "x"), get(child, "y"), get(child, "z")])
        elif tag.lower() == "lookat":
            target = parse_str(child.get("target"))
            origin = parse_str(child.get("origin"))
            look = target - origin
            length = np.linalg.norm(look)
            if length < 1e-5:
                raise ValueError("Normal length too small: Target and origin seems to be the same point")
            rot = look / length
            trans = origin
        else:
            raise ValueError(f"Unsupported transformation representation '{tag}'")
    return rot, trans, scale


def parse_sphere_element(elem: xet.Element):
    """<shape type="sphere"> -> ((1,2,3) [centre; r,r,r], placeholder normal)
    (reference general_parser.py:100-105)."""
    info = np.zeros((1, 2, 3), np.float32)
    info[0, 0] = vec3d_parse(elem.find("point"))
    info[0, 1] = np.full((3,), get(elem.find("float"), "value"))
    return info, np.float32([[0, 1, 0]])
