"""Readers for the leaf values of an AdaPT scene file (format version 1.1): numbers, colours, points, transforms, spheres.

The public names and their return conventions are those of the reference's `parsers/general_parser.py` (`get` :13, `parse_str`
:17, `rgb_parse` :29, `vec3d_parse` :48, `transform_parse` :56, `parse_sphere_element` :100) because scene files, the other
parsers of this package and AdaPT's own callers rely on them.  The numeric flow is kept as well - float32 vectors, float64 scipy
rotation matrices - since the packed geometry must equal, bit for bit, what the reference hands to its kernels
(tests/test_parser.py compares against the reference's parser output).
"""
from __future__ import annotations

import xml.etree.ElementTree as xet
from typing import Callable, Dict, Optional, Tuple

import numpy as np
from scipy.spatial.transform import Rotation

__all__ = ["get", "parse_str", "rgb_parse", "vec3d_parse", "transform_parse", "parse_sphere_element"]

_F32 = np.float32


def get(node: xet.Element, name: str, _type=float):
    """XML attribute -> `_type`; an absent attribute counts as the text "0"."""
    raw = node.get(name)
    return _type("0" if raw is None else raw)


def _xyz(node: xet.Element, keys=("x", "y", "z")) -> np.ndarray:
    return _F32([get(node, k) for k in keys])


def parse_str(val_str: str, no_else_branch: bool = False) -> np.ndarray:
    """Text -> float32 vector.  A comma-separated list wins over a blank-separated one; text holding neither separator is one
    number, repeated three times (grey colour) unless `no_else_branch` forbids the shorthand."""
    sep = "," if "," in val_str else (" " if " " in val_str else None)
    if sep is None:
        if no_else_branch:
            raise ValueError(f"'{val_str}': a vector is required here (components separated by ',' or a blank)")
        return _F32([float(val_str.strip())] * 3)
    return _F32([float(item.strip()) for item in val_str.split(sep)])


def _hex_colour(text: str) -> np.ndarray:
    # "#RRGGBB": each byte / 255 in Python floats, stored as float32
    rgb = np.empty(3, dtype=_F32)
    for channel, start in enumerate((1, 3, 5)):
        rgb[channel] = int(text[start:start + 2], 16) / 255.
    return rgb


def rgb_parse(elem: Optional[xet.Element]) -> np.ndarray:
    """<rgb value="#RRGGBB" | "r, g, b" | "grey"/> or <rgb r=".." g=".." b=".."/> -> float32[3]"""
    if elem is None:
        raise ValueError("colour element expected, got None")
    text = elem.get("value")
    if text is not None:
        return _hex_colour(text) if text.startswith("#") else parse_str(text)
    if not elem.get("r"):
        raise ValueError("colour element carries neither `value` nor `r` / `g` / `b`")
    return _xyz(elem, ("r", "g", "b"))


def vec3d_parse(elem: xet.Element):
    """<point x= y= z=/> -> float32[3] (None for any other tag).  A <value> child switches to the list form, where the
    one-number shorthand is not accepted."""
    if elem.tag != "point":
        return None
    if elem.find("value") is not None:
        return parse_str(elem.get("value"), no_else_branch=True)
    return _xyz(elem)


# ---- <transform> children.  Each reader returns the slots it fills: (rotation, translation, scale)
def _rotation(node: xet.Element):
    kind = node.get("type", "euler")
    if kind == "euler":                    # roll / pitch / yaw in degrees, intrinsic "zxy" order
        return Rotation.from_euler("zxy", tuple(get(node, k) for k in ("r", "p", "y")), degrees=True).as_matrix()
    if kind == "quaternion":
        return Rotation.from_quat([get(node, k) for k in ("x", "y", "z", "w")]).as_matrix()
    if kind == "angle-axis":
        axis = _xyz(node)
        # kept from upstream (general_parser.py:77): the axis is DIVIDED by |axis| * angle in radians, so existing scene files
        # that use this form load exactly as they do there
        axis /= np.linalg.norm(axis) * get(node, "angle") / 180. * np.pi
        return Rotation.from_rotvec(axis).as_matrix()
    raise ValueError(f"rotation type '{kind}' is not one of euler / quaternion / angle-axis")


def _look_at(node: xet.Element):
    """`lookat` fills the rotation slot with the unit viewing DIRECTION (not a matrix) and the translation slot with the eye
    position; `up` is not used."""
    eye = parse_str(node.get("origin"))
    gaze = parse_str(node.get("target")) - eye
    dist = np.linalg.norm(gaze)
    if dist < 1e-5:
        raise ValueError("lookat: target and origin coincide")
    return gaze / dist, eye


def transform_parse(transform_elem: xet.Element) -> Tuple[Optional[np.ndarray], Optional[np.ndarray], Optional[np.ndarray]]:
    """<transform> -> (rotation matrix | look direction, translation | eye position, scale); a slot no child fills is None.
    Later children overwrite earlier ones of the same slot."""
    slots: Dict[str, Optional[np.ndarray]] = {"r": None, "t": None, "s": None}
    for child in transform_elem:
        tag = child.tag
        if tag == "rotate":
            slots["r"] = _rotation(child)
        elif tag == "translate":
            slots["t"] = _xyz(child)
        elif tag == "scale":
            slots["s"] = _xyz(child)
        elif tag.lower() == "lookat":
            slots["r"], slots["t"] = _look_at(child)
        else:
            raise ValueError(f"<{tag}> is not a transform this format knows (translate, rotate, scale, lookat)")
    return slots["r"], slots["t"], slots["s"]


def parse_sphere_element(elem: xet.Element):
    """<shape type="sphere"> -> (float32 (1, 2, 3): row 0 the centre, row 1 the radius three times; a placeholder normal)"""
    radius = get(elem.find("float"), "value")
    packed = np.zeros((1, 2, 3), _F32)
    packed[0, 0] = vec3d_parse(elem.find("point"))
    packed[0, 1] = radius
    return packed, _F32([[0, 1, 0]])
