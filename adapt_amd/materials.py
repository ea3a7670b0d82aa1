"""Host-side material records (XML -> parameters) for the `pt` hot path.

Mirrors the *host* halves of the reference's `bxdf/brdf.py:35-140` (BRDF_np),
`bxdf/bsdf.py:29-58` (BSDF_np) and `bxdf/medium.py:24-69` (Medium_np): same
class names, same attribute names (`type_id`, `k_d`, `k_s`, `k_g`, `is_delta`,
`medium.ior`), same derived parameters.  The device halves live in
`csrc/shading.hpp` (HIP) — there is no Taichi struct to export to; `pack()`
returns the flat float/int record the C-ABI scene description wants.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as xet

import numpy as np

from .parsers.general_parser import get, rgb_parse

__all__ = ["BRDFTag", "BRDF_np", "BSDF_np", "Medium_np"]

DEG2RAD = np.pi / 180.

# The reference's module switch `__ENABLE_MICROFACET__` (bxdf/brdf.py:8), off upstream "since microfacet functions can slow down JIT
# compilation".  Off: a microfacet BRDF parses to a Lambertian one, as upstream (brdf.py:60-65).  On (set it here, or
# ADAPT_ENABLE_MICROFACET=1 in the environment): BRDF type 3 reaches the device and shades as upstream's Trowbridge-Reitz model
# (sampler/microfacet.py, brdf.py:428-484) - there is nothing to compile at run time here, so the switch costs nothing.
ENABLE_MICROFACET = os.environ.get("ADAPT_ENABLE_MICROFACET", "0") == "1"


class BRDFTag:
    """Model ids, reference renderer/constants.py:44-52."""
    BLING_PHONG = 0
    LAMBERTIAN = 1
    SPECULAR = 2
    MICROFACET = 3
    MOD_PHONG = 4
    FRESNEL_BLEND = 5
    OREN_NAYAR = 6
    THIN_COAT = 7


_ALBEDO_KEYS = frozenset(("reflectance", "albedo", "k_d"))
_GLOSS_KEYS = frozenset(("glossiness", "shininess", "roughness", "sigma", "k_g"))
_SPEC_KEYS = frozenset(("specular", "ref_ior", "k_s"))
_BRDF_IDS = {"phong": 0, "lambertian": 1, "specular": 2, "microfacet": 3,
             "mod-phong": 4, "fresnel-blend": 5, "oren-nayar": 6, "thin-coat": 7}
_BSDF_IDS = {"det-refraction": 0, "null": -1, "lambertian": 1}
_MEDIUM_IDS = {"hg": 0, "multi-hg": 1, "rayleigh": 2, "mie": 3, "transparent": -1}


def _roughness_to_alpha(r: np.ndarray) -> np.ndarray:
    # PBRT-v3 TrowbridgeReitzDistribution::RoughnessToAlpha polynomial (brdf.py:115-120)
    x = np.log(np.maximum(r, 1e-3))
    return 1.62142 + 0.819955 * x + 0.1734 * x * x + 0.0171201 * (x ** 3) + 0.000640711 * (x ** 4)


class BRDF_np:
    """<brdf type=... id=...> with <rgb name=k_d|k_s|k_g (+aliases)> children."""

    is_bsdf = False

    def __init__(self, elem: xet.Element, no_setup: bool = False):
        self.type = elem.get("type")
        self.type_id = _BRDF_IDS.get(self.type, -1)
        self.id = elem.get("id")
        self.k_d = np.ones(3, np.float32)
        self.k_s = np.zeros(3, np.float32)
        self.k_g = np.ones(3, np.float32)
        self.is_delta = False
        if self.type_id == BRDFTag.MICROFACET and not ENABLE_MICROFACET:
            # microfacet is switched off upstream (brdf.py:8,60-65): Lambertian fallback
            self.type, self.type_id = "lambertian", BRDFTag.LAMBERTIAN
        for node in elem.findall("rgb"):
            name = node.get("name")
            if name is None:
                raise ValueError(f"RGB node in BR(S)DF <{self.id}> has empty name.")
            if name in _ALBEDO_KEYS:
                self.k_d = rgb_parse(node)
            elif name in _SPEC_KEYS:
                self.k_s = rgb_parse(node)
            elif name in _GLOSS_KEYS:
                self.k_g = rgb_parse(node)
                if name == "roughness":
                    self.k_g = _roughness_to_alpha(self.k_g.clip(0, 1))
                elif name == "sigma":
                    # Oren-Nayar A/B from the spherical-gaussian sigma (brdf.py:104-110)
                    s2 = (self.k_g[0] * DEG2RAD) ** 2
                    self.k_g[0] = 1 - (s2 / (2 * (s2 + 0.33)))
                    self.k_g[1] = 0.45 * s2 / (s2 + 0.09)
                    self.k_g[2] = max(1., self.k_g[2])
        if not no_setup:
            self.setup()

    def setup(self):
        if self.type not in _BRDF_IDS:
            raise NotImplementedError(f"Unknown BRDF type: {self.type}")
        if self.type_id == BRDFTag.SPECULAR:
            self.is_delta = True
        elif self.type_id == BRDFTag.FRESNEL_BLEND:
            # sqrt((nu+1)(nv+1)) / 8pi cached in k_g[2] (brdf.py:127-128)
            self.k_g[2] = np.sqrt((self.k_g[0] + 1) * (self.k_g[1] + 1)) / (8. * np.pi)

    @property
    def mean(self) -> np.ndarray:
        """(mean k_d, mean k_s, mean k_g) — `BRDF.mean`, brdf.py:136."""
        return np.float32([self.k_d.mean(), self.k_s.mean(), self.k_g.mean()])

    def pack(self):
        """-> (int32[4] = type, is_delta, is_bsdf, 0 ; float32[13] = k_d,k_s,k_g,mean,ior)."""
        if self.type_id == -1 and not self.is_bsdf:
            raise ValueError("BRDF not properly initialised (type_id = -1)")
        ints = np.int32([self.type_id, int(self.is_delta), int(self.is_bsdf), 0])
        ior = np.float32(self.medium.ior) if self.is_bsdf else np.float32(1.0)
        flts = np.concatenate([self.k_d, self.k_s, self.k_g, self.mean, [ior]]).astype(np.float32)
        return ints, flts

    def __repr__(self):
        return f"<{self.type.capitalize()} BRDF id={self.id}>"


class Medium_np:
    """<medium type=...> ; for the pt path only `ior` matters (bsdf.py:87-88)."""

    def __init__(self, elem: xet.Element | None, is_world: bool = False):
        self.ior = 1.0
        self.u_a = np.zeros(3, np.float32)
        self.u_s = np.zeros(3, np.float32)
        self.par = np.zeros(3, np.float32)
        self.pdf = np.float32([1., 0., 0.])
        self.type_id = -1
        self.type_name = "transparent"
        if elem is not None:
            kind = elem.get("type")
            if kind not in _MEDIUM_IDS:
                raise NotImplementedError(f"Medium type '{kind}' is not supported.")
            self.type_id, self.type_name = _MEDIUM_IDS[kind], kind
            for node in elem.findall("rgb"):
                if hasattr(self, node.get("name")):
                    setattr(self, node.get("name"), rgb_parse(node))
            for node in elem.findall("float"):
                if hasattr(self, node.get("name")):
                    setattr(self, node.get("name"), get(node, "value"))
        self.u_e = self.u_a + self.u_s

    def __repr__(self):
        return f"<Medium {self.type_name.capitalize()} ior {self.ior:.3f}>"


class BSDF_np(BRDF_np):
    """<bsdf type=det-refraction|lambertian|null> with an attached <medium>."""

    is_bsdf = True

    def __init__(self, elem: xet.Element):
        super().__init__(elem, True)
        self.medium = Medium_np(elem.find("medium"))
        if self.type not in _BSDF_IDS:
            raise NotImplementedError(f"Unknown BSDF type: {self.type}")
        self.type_id = _BSDF_IDS[self.type]
        self.is_delta = self.type_id == 0          # bsdf.py:37-41

    def setup(self):
        pass

    def __repr__(self):
        return f"<{self.type.capitalize()} BSDF with {self.medium!r}>"
