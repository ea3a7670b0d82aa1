"""taichi.math stand-in (see taichi/__init__.py)."""
import math as _pm
from . import (_TensorType, Vector, Matrix, dot, cross, pow, sign, max, min, cos, sin, tan, floor, acos, atan2, mix,
               isnan, isinf, sqrt, exp, log)

pi = _pm.pi
vec2 = _TensorType((2,))
vec3 = _TensorType((3,))
vec4 = _TensorType((4,))
mat3 = _TensorType((3, 3))
ivec3 = _TensorType((3,), int)


def clamp(x, xmin, xmax):
    """taichi.math.clamp: min(xmax, max(xmin, x))"""
    return min(xmax, max(xmin, x))
