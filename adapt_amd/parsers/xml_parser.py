"""Scene XML (AdaPT format v1.1) -> the four values the renderer constructor takes.

Drop-in counterpart of reference `parsers/xml_parser.py:246-289`:

    emitter_configs, array_info, all_objs, configs = scene_parsing(directory, file)

* `emitter_configs` list of LightSource (area emitters get `inv_area` from the
  shape they are attached to, xml_parser.py:56-64)
* `array_info`     {"primitives" (N,3,3) f32, "indices" int64[] | None (which
  rows are spheres), "n_g" (N,3), "n_s" (N,3,3), "uvs" (N,3,2)}
* `all_objs`       list of ObjDescriptor
* `configs`        sensor dict: every <integer|float|string|boolean name=..>
  child, + "transform" (look dir, origin, None), "film" {...}, "world",
  "packed_textures" ({tag: atlas | None} or None), "has_vertex_normal".

Image textures (`<texture type="image">`, albedo / normal / bump maps on meshes) are read by adapt_amd/textures.py
with its own atlas packer; checkerboard textures (no lookup upstream) and textured spheres are refused.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as xet
from typing import Dict, List

import numpy as np

from ..emitters import SOURCE_MAP, LightSource
from ..materials import BRDF_np, BSDF_np
from ..textures import TEX_TAGS, parse_texture
from .general_parser import get, parse_sphere_element, transform_parse
from .obj_desc import ObjDescriptor
from .obj_loader import SPHERE, TRIANGLE_MESH, apply_transform, calculate_surface_area, extract_obj_info
from .world import World_np

__all__ = ["scene_parsing", "SCENE_VERSION"]

SCENE_VERSION = "1.1"
_CASTS = {"integer": int, "float": float, "string": str, "boolean": lambda s: s.lower() == "true"}


def _zeros_if_none(arr, n_prims: int, last: int = 3):
    return np.zeros((n_prims, 3, last), dtype=np.float32) if arr is None else arr


def parse_emitters(nodes: List[xet.Element]):
    sources: List[LightSource] = []
    index_of: Dict[str, int] = {}
    for node in nodes:
        kind = node.get("type")
        cls = SOURCE_MAP.get(kind)
        if cls is None:
            raise ValueError(f"Source type '{kind}' is not supported. Please check your XML settings.")
        src = cls(node)
        if src.id in index_of:
            raise ValueError(f"Two sources with same id {src.id} will result in conflicts")
        index_of[src.id] = len(sources)
        sources.append(src)
    return sources, index_of


def parse_bxdf(nodes: List[xet.Element]):
    table = {}
    for node in nodes:
        table[node.get("id")] = BRDF_np(node) if node.tag == "brdf" else BSDF_np(node)
    return table


def parse_wavefront(directory: str, shapes: List[xet.Element], bsdf_dict: dict, emitter_dict: dict, texture_dict=None):
    """Shapes in document order -> packed arrays + descriptors (xml_parser.py:93-176)."""
    objs, prims, uvs_all, ng_all, ns_all, sphere_rows = [], [], [], [], [], []
    area_of_emitter = {}
    has_vn = False
    row = 0
    for shape in shapes:
        vns = uvs = rot = trans = None
        kind = TRIANGLE_MESH
        if shape.get("type") == "obj":
            rel = shape.find("string").get("value")
            meshes, normals, vns, uvs = extract_obj_info(os.path.join(directory, rel))
            tnode = shape.find("transform")
            if tnode is not None:
                rot, trans, scl = transform_parse(tnode)
                meshes, normals = apply_transform(meshes, normals, rot, trans, scl)
            has_vn = has_vn or vns is not None
        else:
            meshes, normals = parse_sphere_element(shape)
            kind = SPHERE
        material = None
        emit_id = -1
        group = {t: None for t in TEX_TAGS}
        for ref in shape.findall("ref"):
            rtype, rid = ref.get("type"), ref.get("id")
            if rtype == "material":
                material = bsdf_dict[rid]
            elif rtype == "emitter":
                emit_id = emitter_dict[rid]
                area_of_emitter[emit_id] = calculate_surface_area(meshes, kind)
            elif rtype == "texture":
                tag = ref.get("tag", None)
                if tag is None or tag not in group:
                    tag = "albedo"                                   # xml_parser.py:142-147
                if texture_dict is None or texture_dict.get(tag) is None or rid not in texture_dict[tag]:
                    raise KeyError(f"Texture id '{rid}' does not have tag '{tag}' mapping, check if it is from other groups.")
                if kind == SPHERE:
                    raise NotImplementedError("textured spheres: upstream looks the texture up with whatever (u, v) the last accepted "
                                              "triangle left behind (tracer_base.py:184-237); not reproduced")
                group[tag] = texture_dict[tag][rid]
        if material is None:
            raise ValueError("Object should be attached with a BSDF for now since no default one implemented yet.")
        n = meshes.shape[0]
        if kind == SPHERE:          # (1,2,3) -> (1,3,3): centre, (r,r,r), 0
            meshes = np.concatenate((meshes, np.zeros((1, 1, 3), dtype=np.float32)), axis=-2)
            sphere_rows.append(row)
        prims.append(meshes)
        ng_all.append(normals)
        ns_all.append(_zeros_if_none(vns, n))
        uvs_all.append(_zeros_if_none(uvs, n, 2))
        objs.append(ObjDescriptor(meshes, normals, material, vns, uvs, group, rot, trans, emit_id, kind))
        row += n
    array_info = {
        "primitives": np.concatenate(prims, axis=0).astype(np.float32),
        "indices": np.int64(sphere_rows) if sphere_rows else None,
        "n_g": np.concatenate(ng_all, axis=0).astype(np.float32),
        "n_s": np.concatenate(ns_all, axis=0).astype(np.float32),
        "uvs": np.concatenate(uvs_all, axis=0).astype(np.float32),
    }
    return array_info, objs, area_of_emitter, has_vn


def parse_global_sensor(sensor: xet.Element) -> dict:
    """Typed children by tag; later duplicates win (xml_parser.py:225-244)."""
    cfg = {}
    for child in sensor:
        if child.tag in _CASTS:
            cfg[child.get("name")] = get(child, "value", _CASTS[child.tag])
    cfg["transform"] = transform_parse(sensor.find("transform"))
    film_ints = sensor.find("film").findall("integer")
    assert len(film_ints) >= 2
    cfg["film"] = {e.get("name"): get(e, "value", int) for e in film_ints}
    return cfg


def scene_parsing(directory: str, file: str):
    root = xet.parse(os.path.join(directory, file)).getroot()
    version = root.attrib["version"]
    if version != SCENE_VERSION:
        raise ValueError(f"Unsupported version {version}. Only '{SCENE_VERSION}' is supported right now.")
    sensor = root.find("sensor")
    assert sensor is not None
    emitters, emitter_dict = parse_emitters(root.findall("emitter"))
    bsdf_dict = parse_bxdf(root.findall("bsdf") + root.findall("brdf"))
    teximgs, textures = parse_texture(root.findall("texture"), base_dir=directory)
    array_info, objs, areas, has_vn = parse_wavefront(directory, root.findall("shape"), bsdf_dict, emitter_dict, textures)
    cfg = parse_global_sensor(sensor)
    cfg["world"] = World_np(root.find("world"))
    cfg["packed_textures"] = teximgs
    cfg["has_vertex_normal"] = has_vn
    cfg["volume"] = root.findall("volume")[:1]
    for i, em in enumerate(emitters):
        if i in areas:
            em.inv_area = 1. / areas[i]
            em.attached = True
        elif em.type == "area":
            raise ValueError("Setting L1 / L2 for area light is deprecated a long ago. Please attach area light to an object.")
    return emitters, array_info, objs, cfg
