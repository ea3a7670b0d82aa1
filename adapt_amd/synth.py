"""Deterministic stand-ins for the two BASELINE scenes whose assets the reference does not ship.

`README.md:56` ("Bunny scenes are not uploaded") and `scenes/.gitignore`: the three-bunnies
meshes/textures referenced by `scenes/cbox/bunny.xml:155-185` and the sports-car scene are
absent from the reference tree.  These generators build scenes of the same
size class from what IS bundled — the Cornell room quads and `meshes/cornell/bunny.obj`
(495 triangles), midpoint-subdivided three times (31 680 triangles per instance) — with the
materials, lights and sensor settings of `bunny.xml` (C4) / a mixed-BxDF set (C5):

    three_bunnies()  : room (10 tris) + 3 bunnies  =  95 050 triangles, 3 spot lights, S=2, 8 bounces, 800x800   (C4)
    bunny_field()    : room + 2 luminaires + 9 bunnies = 285 134 triangles, 2 area lights, all 7 live BRDFs + glass,
                       S=1, 16 bounces, 1280x720                                                                   (C5)

No RNG is involved: both the CPU oracle and the HIP path are handed identical arrays.  The
return value is the 4-tuple `scene_parsing` returns, so everything downstream is unchanged.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as xet
from typing import List, Tuple

import numpy as np

from .emitters import SOURCE_MAP
from .materials import BRDF_np, BSDF_np
from .parsers.obj_desc import ObjDescriptor
from .parsers.obj_loader import calculate_surface_area, extract_obj_info
from .parsers.world import World_np

__all__ = ["three_bunnies", "bunny_field", "subdivide", "SYNTH_SCENES"]

_MESH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes", "meshes", "cornell")


def subdivide(tris: np.ndarray, levels: int) -> np.ndarray:
    """Midpoint subdivision, float32: (N,3,3) -> (N*4^levels,3,3); child order (a,ab,ca) (ab,b,bc) (ca,bc,c) (ab,bc,ca)."""
    t = np.ascontiguousarray(tris, np.float32)
    half = np.float32(0.5)
    for _ in range(levels):
        a, b, c = t[:, 0], t[:, 1], t[:, 2]
        ab, bc, ca = (a + b) * half, (b + c) * half, (c + a) * half
        t = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], 1).reshape(-1, 3, 3)
    return np.ascontiguousarray(t, np.float32)


def _geo_normals(m: np.ndarray) -> np.ndarray:
    n = np.cross(m[:, 1] - m[:, 0], m[:, 2] - m[:, 1])
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    return n.astype(np.float32)


def _bunny(levels: int) -> np.ndarray:
    m, _, _, _ = extract_obj_info(os.path.join(_MESH_DIR, "bunny.obj"))
    return subdivide(m, levels)


def _place(mesh: np.ndarray, scale: float, foot: Tuple[float, float, float]) -> np.ndarray:
    """Uniformly scale about the bounding-box bottom centre and put that point at `foot` (float32 arithmetic)."""
    lo, hi = mesh.min(axis=(0, 1)), mesh.max(axis=(0, 1))
    anchor = np.float32([(lo[0] + hi[0]) * 0.5, lo[1], (lo[2] + hi[2]) * 0.5])
    return ((mesh - anchor) * np.float32(scale) + np.float32(foot)).astype(np.float32)


def _mat(xml: str):
    node = xet.fromstring(xml)
    return BRDF_np(node) if node.tag == "brdf" else BSDF_np(node)


def _brdf(kind, kd, kg="1.0", ks="0.0"):
    return _mat(f'<brdf type="{kind}" id="{kind}"><rgb name="k_d" value="{kd}"/><rgb name="k_g" value="{kg}"/><rgb name="k_s" value="{ks}"/></brdf>')


def _glass(kd="#FFFFFF", ior=1.5):
    return _mat(f'<bsdf type="det-refraction" id="glass"><rgb name="k_d" value="{kd}"/><medium type="transparent"><float name="ior" value="{ior}"/></medium></bsdf>')


class _Builder:
    def __init__(self):
        self.objs: List[ObjDescriptor] = []
        self.prims, self.ng, self.ns, self.uvs = [], [], [], []
        self.area_of = {}
        self.sphere_rows: List[int] = []

    def sphere(self, centre, radius, material):
        """one analytic sphere (packed as the parser packs it: centre, (r, r, r), 0 - xml_parser.parse_wavefront)"""
        rec = np.zeros((1, 3, 3), np.float32)
        rec[0, 0] = np.float32(centre); rec[0, 1] = np.float32(radius)
        self.sphere_rows.append(sum(p.shape[0] for p in self.prims))
        self.prims.append(rec); self.ng.append(np.float32([[0, 1, 0]])); self.ns.append(np.zeros((1, 3, 3), np.float32)); self.uvs.append(np.zeros((1, 3, 2), np.float32))
        self.objs.append(ObjDescriptor(rec, np.float32([[0, 1, 0]]), material, None, None, {"albedo": None, "normal": None, "bump": None, "roughness": None},
                                       None, None, -1, 1))

    def mesh(self, tris, material, vns=None, emitter=-1):
        tris = np.ascontiguousarray(tris, np.float32)
        ng = _geo_normals(tris)
        if vns is None:
            vns = np.repeat(ng[:, None, :], 3, axis=1)            # flat shading normals (bunny.obj ships per-face vn)
        self.prims.append(tris); self.ng.append(ng); self.ns.append(np.float32(vns)); self.uvs.append(np.zeros((tris.shape[0], 3, 2), np.float32))
        if emitter >= 0:
            self.area_of[emitter] = calculate_surface_area(tris)
        self.objs.append(ObjDescriptor(tris, ng, material, vns, None, {"albedo": None, "normal": None, "bump": None, "roughness": None},
                                       None, None, emitter, 0))

    def cornell(self, name, material, translate=None, emitter=-1):
        m, n, vn, _ = extract_obj_info(os.path.join(_MESH_DIR, f"cbox_{name}.obj"))
        if translate is not None:
            m = m + np.float32(translate)
        self.mesh(m, material, vn, emitter)

    def finish(self, emitters, sensor_xml: str):
        cfg = {}
        sensor = xet.fromstring(sensor_xml)
        casts = {"integer": int, "float": float, "string": str, "boolean": lambda s: s.lower() == "true"}
        for ch in sensor:
            if ch.tag in casts:
                cfg[ch.get("name")] = casts[ch.tag](ch.get("value"))
        cfg["transform"] = (np.float32([0, 0, 1]), np.float32([2.78, 2.73, -8.0]), None)     # lookat of every bundled Cornell scene
        cfg["film"] = {"width": cfg.pop("width"), "height": cfg.pop("height")}
        cfg["world"] = World_np(None)
        cfg["packed_textures"] = None
        cfg["has_vertex_normal"] = True
        cfg["volume"] = []
        for i, em in enumerate(emitters):
            if i in self.area_of:
                em.inv_area = 1. / self.area_of[i]
                em.attached = True
        arr = {"primitives": np.concatenate(self.prims).astype(np.float32), "indices": np.int64(self.sphere_rows) if self.sphere_rows else None,
               "n_g": np.concatenate(self.ng).astype(np.float32), "n_s": np.concatenate(self.ns).astype(np.float32),
               "uvs": np.concatenate(self.uvs).astype(np.float32)}
        return emitters, arr, self.objs, cfg


def _sensor(w, h, bounce, nshadow):
    return (f'<sensor><float name="fov" value="39.3077"/><integer name="max_bounce" value="{bounce}"/><integer name="num_shadow_ray" value="{nshadow}"/>'
            f'<boolean name="use_rr" value="true"/><boolean name="anti_alias" value="true"/><boolean name="stratified_sampling" value="true"/>'
            f'<boolean name="use_mis" value="true"/><string name="accelerator" value="bvh"/><integer name="width" value="{w}"/>'
            f'<integer name="height" value="{h}"/></sensor>')


def _room(b: _Builder, white, left, right):
    b.cornell("floor", white); b.cornell("ceiling", white); b.cornell("back", white)
    b.cornell("greenwall", right); b.cornell("redwall", left)


def _spot(emission, scaler, pos, direc, half_angle, ident):
    xml = (f'<emitter type="spot" id="{ident}"><rgb name="emission" value="{emission}"/><rgb name="scaler" value="{scaler}"/>'
           f'<point name="pos" x="{pos[0]}" y="{pos[1]}" z="{pos[2]}"/><point name="dir" x="{direc[0]}" y="{direc[1]}" z="{direc[2]}"/>'
           f'<float name="half-angle" value="{half_angle}"/></emitter>')
    return SOURCE_MAP["spot"](xet.fromstring(xml))


def three_bunnies(levels: int = 3):
    """C4 stand-in (materials / lights / sensor of scenes/cbox/bunny.xml:8-22,76-98,130-152)."""
    b = _Builder()
    white, left, right = _brdf("lambertian", "#BDBDBD"), _brdf("lambertian", "#DD2525"), _brdf("lambertian", "#25DD25")
    _room(b, white, left, right)
    bunny = _bunny(levels)
    lava = _brdf("lambertian", "#FFFFFF")
    fresnel = _mat('<brdf type="fresnel-blend" id="fresnel"><rgb name="k_d" value="#CACACA"/><rgb name="k_s" value="#333333"/><rgb name="k_g" r="10" g="1000"/></brdf>')
    for mat, foot in ((lava, (4.2, 0.0, 3.9)), (_glass(), (2.75, 0.0, 2.3)), (fresnel, (1.25, 0.0, 3.7))):
        b.mesh(_place(bunny, 0.45, foot), mat)
    emitters = [_spot("6.0, 4.0, 4.0", "245.0", (3.779, 5.2, 2.745), (-0.2, -1.5, -0.3), 20.0, "source1"),
                _spot("6.0, 6.0, 4.0", "200.0", (1.2, 4.8, 3.2), (0.6, -1.5, -0.05), 15.0, "source2"),
                _spot("4.0, 4.0, 6.0", "200.0", (4.9, 2.5, 3.8), (-1.6, -0.6, -0.4), 15.0, "source3")]
    return b.finish(emitters, _sensor(800, 800, 8, 2))


def bunny_field(levels: int = 3):
    """C5 stand-in: a 3x3 field of bunnies cycling through every live surface model, two area lights."""
    b = _Builder()
    white, left, right = _brdf("phong", "#BDBDBD"), _brdf("phong", "#DD2525"), _brdf("phong", "#25DD25")
    area = [SOURCE_MAP["area"](xet.fromstring(f'<emitter type="area" id="a{k}"><rgb name="emission" value="{e}"/></emitter>'))
            for k, e in enumerate(("50.0, 45.6, 42.3", "30.0, 32.0, 40.0"))]
    light = _brdf("phong", "#555555")
    b.cornell("luminaire", light, (0.9, -0.001, 0.6), emitter=0)
    b.cornell("luminaire", light, (-1.0, -0.001, -0.9), emitter=1)
    _room(b, white, left, right)
    bunny = _bunny(levels)
    mats = [_brdf("phong", "#BCBCBC", "8.0", "#303030"), _brdf("lambertian", "#FFFFFF"), _brdf("specular", "#DEDEDE"),
            _brdf("mod-phong", "#BCBCBC", "10.0", "#424242"),
            _mat('<brdf type="fresnel-blend" id="fb"><rgb name="k_d" value="#CACACA"/><rgb name="k_s" value="#333333"/><rgb name="k_g" r="10" g="1000"/></brdf>'),
            _mat('<brdf type="oren-nayar" id="on"><rgb name="k_d" value="#C8B496"/><rgb name="sigma" value="20.0"/></brdf>'),
            _mat('<brdf type="thin-coat" id="tc"><rgb name="k_d" value="#9696C8"/><rgb name="k_s" value="0.9"/><rgb name="sigma" r="20" g="20" b="1.5"/></brdf>'),
            _glass("#FAFAFA"), _brdf("lambertian", "#E0C080")]
    k = 0
    for zi in range(3):
        for xi in range(3):
            b.mesh(_place(bunny, 0.3, (0.95 + 1.8 * xi, 0.0, 0.9 + 1.75 * zi)), mats[k])
            k += 1
    return b.finish(area, _sensor(1280, 720, 16, 1))


SYNTH_SCENES = {"three-bunnies": three_bunnies, "bunny-field": bunny_field}
