"""Stand-in for the absent third-party `taichi==1.6.0` package — GOLDEN GENERATION ONLY.

Used by tests/golden/gen/gen_goldens.py in the authoring container to execute the
reference's own, unmodified Python source (/root/reference) in Python scope so that
fixtures can be recorded from it.  It never runs on the GPU box and is not imported
by the product or by the tests themselves.

Semantics emulated (what the pt path relies on):
  * default_fp = f32: every scalar/vector/matrix value is numpy float32; Python
    literals are weak (NumPy 2 promotion), transcendental functions go through
    glibc's float entry points (cosf, powf, ...), i.e. the same libm the C oracle uses
  * value semantics: reads of vector members / field elements copy; augmented
    assignment rebinds instead of mutating
  * vector ops as published in taichi/lang/matrix_ops.py: sum/dot accumulate left
    to right, normalized(v) = (1/norm(v)) * v, 3x3 inverse = adjugate * (1/det)
  * ti.select evaluates both branches; integer % is Python's
  * fields / SNodes (dense, bitmasked + is_active), struct dataclasses with .field()
  * ti.random: either a scripted list or the counter-based Philox-4x32-10 stream
    (key = (pixel, seed), counter = (sample, draw // 4)) shared with oracle/ and the HIP path
"""
import ctypes as _C
import numpy as _np

class f32(_np.float32):
    """The scalar the reference code computes with.  A numpy float32 whose arithmetic stays in this class, so that the two places
    where numpy's own scalar semantics differ from Taichi's can be put right:
      * `x ** n` with a small constant integer n is a multiplication chain (Taichi's simplifier; numpy calls powf, <= 1 ulp off),
      * float `a % b` is a - b * floor(a / b) (taichi/lang/ops.py mod; numpy's remainder differs when a / b rounds up to an integer)."""
    __slots__ = ()

    def _wrap(r):
        return f32(r) if isinstance(r, _np.floating) else r

    def __add__(s, o): return f32._wrap(_np.float32.__add__(s, o))
    def __radd__(s, o): return f32._wrap(_np.float32.__radd__(s, o))
    def __sub__(s, o): return f32._wrap(_np.float32.__sub__(s, o))
    def __rsub__(s, o): return f32._wrap(_np.float32.__rsub__(s, o))
    def __mul__(s, o): return f32._wrap(_np.float32.__mul__(s, o))
    def __rmul__(s, o): return f32._wrap(_np.float32.__rmul__(s, o))
    def __truediv__(s, o): return f32._wrap(_np.float32.__truediv__(s, o))
    def __rtruediv__(s, o): return f32._wrap(_np.float32.__rtruediv__(s, o))
    def __neg__(s): return f32(_np.float32.__neg__(s))
    def __pos__(s): return s
    def __abs__(s): return f32(_np.float32.__abs__(s))

    def __pow__(s, o):
        if isinstance(o, Tensor):
            return NotImplemented
        return pow(s, o)

    def __rpow__(s, o):
        return pow(f32(o), s)

    def __mod__(s, o):
        if isinstance(o, Tensor):
            return NotImplemented
        o = f32(o)
        return f32(s - o * f32(_np.floor(_np.float32(s) / _np.float32(o))))

    def __rmod__(s, o):
        return f32(o).__mod__(s)


i32 = int32 = int
float32 = f32
i8 = u8 = i16 = u32 = u64 = i64 = int
f64 = f32
cpu, cuda, gpu, vulkan = "cpu", "cuda", "gpu", "vulkan"
i, j, k = 0, 1, 2
ij = (0, 1)

_libm = _C.CDLL("libm.so.6")
for _n in ("cosf", "sinf", "tanf", "acosf", "expf", "logf", "floorf"):
    getattr(_libm, _n).restype = _C.c_float
    getattr(_libm, _n).argtypes = [_C.c_float]
for _n in ("powf", "atan2f"):
    getattr(_libm, _n).restype = _C.c_float
    getattr(_libm, _n).argtypes = [_C.c_float, _C.c_float]


def _m1(name):
    fn = getattr(_libm, name)
    def g(x):
        if isinstance(x, Tensor):
            return type(x)._wrap(_np.array([fn(float(v)) for v in x.a.reshape(-1)], dtype=f32).reshape(x.a.shape))
        return f32(fn(float(f32(x))))
    return g


def _m2(name):
    fn = getattr(_libm, name)
    def g(x, y):
        if isinstance(x, Tensor) or isinstance(y, Tensor):
            xa = x.a if isinstance(x, Tensor) else _np.full(y.a.shape, f32(x), f32)
            ya = y.a if isinstance(y, Tensor) else _np.full(x.a.shape, f32(y), f32)
            cls = type(x) if isinstance(x, Tensor) else type(y)
            out = _np.array([fn(float(a), float(b)) for a, b in zip(xa.reshape(-1), ya.reshape(-1))], dtype=f32)
            return cls._wrap(out.reshape(xa.shape))
        return f32(fn(float(f32(x)), float(f32(y))))
    return g


# --------------------------------------------------------------------- tensors
def _raw(o):
    if isinstance(o, Tensor):
        return o.a
    if isinstance(o, (_np.floating, float)):
        return f32(o)
    return o


class Tensor:
    __array_priority__ = 1000
    __slots__ = ("a",)

    def __init__(self, *args):
        if len(args) == 1:
            d = args[0]
            if isinstance(d, Tensor):
                d = d.a
            self.a = _np.array([_raw(x) for x in d] if isinstance(d, (list, tuple)) else d, dtype=f32).copy()
        else:
            self.a = _np.array([_raw(x) for x in args], dtype=f32)

    @classmethod
    def _wrap(cls, arr):
        o = cls.__new__(cls)
        o.a = _np.asarray(arr, dtype=f32)
        return o

    def copy(self):
        return type(self)._wrap(self.a.copy())

    def _bin(self, other, op, rev=False):
        o = _raw(other)
        if isinstance(o, _np.ndarray) and o.dtype != f32:
            o = o.astype(f32)
        r = op(o, self.a) if rev else op(self.a, o)
        cls = type(self)
        if isinstance(other, Tensor) and other.a.ndim > self.a.ndim:
            cls = type(other)
        return cls._wrap(r.astype(f32))

    def __add__(s, o): return s._bin(o, _np.add)
    def __radd__(s, o): return s._bin(o, _np.add, True)
    def __sub__(s, o): return s._bin(o, _np.subtract)
    def __rsub__(s, o): return s._bin(o, _np.subtract, True)
    def __mul__(s, o): return s._bin(o, _np.multiply)
    def __rmul__(s, o): return s._bin(o, _np.multiply, True)
    def __truediv__(s, o): return s._bin(o, _np.divide)
    def __rtruediv__(s, o): return s._bin(o, _np.divide, True)
    def __pow__(s, o):
        if isinstance(o, int) and o == 2:
            return s * s
        return pow(s, o)
    # augmented ops rebind (value semantics)
    __iadd__, __isub__, __imul__, __itruediv__ = __add__, __sub__, __mul__, __truediv__
    def __neg__(s): return type(s)._wrap(-s.a)
    def __lt__(s, o): return s.a < _raw(o)
    def __le__(s, o): return s.a <= _raw(o)
    def __gt__(s, o): return s.a > _raw(o)
    def __ge__(s, o): return s.a >= _raw(o)
    def __len__(s): return s.a.shape[0]

    def __getitem__(s, idx):
        v = s.a[idx]
        return f32(v) if _np.ndim(v) == 0 else Vector._wrap(v.copy())

    def __setitem__(s, idx, v):
        s.a[idx] = _raw(v)

    def __iter__(s):
        for q in range(s.a.shape[0]):
            yield s[q]

    def fill(s, v): s.a[...] = f32(v)
    def to_numpy(s): return s.a.copy()
    def __repr__(s): return f"{type(s).__name__}({s.a.tolist()})"
    def __array__(s, dtype=None, copy=None): return s.a.astype(dtype) if dtype is not None else s.a.copy()

    def sum(s):
        acc = None
        for v in s.a.reshape(-1):
            acc = v if acc is None else f32(acc + v)
        return f32(acc)

    def max(s): return f32(s.a.max())
    def min(s): return f32(s.a.min())


class Vector(Tensor):
    __slots__ = ()

    def norm_sqr(s): return (s * s).sum()
    def norm(s, eps=0): return f32(_np.sqrt(f32(s.norm_sqr() + f32(eps))))
    def normalized(s, eps=0):
        invlen = f32(1.0) / f32(s.norm() + f32(eps))
        return invlen * s
    def dot(s, o): return (s * o).sum()
    def cross(s, o): return cross(s, o)

    @staticmethod
    def field(n, dtype, shape=None):
        return Field(dtype, (n,), shape)


class Matrix(Tensor):
    __slots__ = ()

    def __init__(self, *args):
        if len(args) == 1:
            d = args[0]
            if isinstance(d, Tensor):
                d = d.a
            if isinstance(d, (list, tuple)):
                d = [[_raw(x) for x in row] for row in d]
            self.a = _np.array(d, dtype=f32).copy()
        else:
            raise TypeError("Matrix(rows)")

    @staticmethod
    def cols(cs):
        return Matrix._wrap(_np.stack([_raw(c) for c in cs], axis=1).astype(f32))

    @staticmethod
    def rows(rs):
        return Matrix._wrap(_np.stack([_raw(c) for c in rs], axis=0).astype(f32))

    @staticmethod
    def zero(dt, n, m=None):
        return Matrix._wrap(_np.zeros((n, m if m is not None else n), f32))

    @staticmethod
    def diag(n, v):
        a = _np.zeros((n, n), f32)
        for q in range(n):
            a[q, q] = f32(v)
        return Matrix._wrap(a)

    def __matmul__(s, o):
        A = s.a
        B = o.a if isinstance(o, Tensor) else _np.asarray(o, f32)
        vec = B.ndim == 1
        if vec:
            B = B.reshape(-1, 1)
        out = _np.zeros((A.shape[0], B.shape[1]), f32)
        for r in range(A.shape[0]):
            for c in range(B.shape[1]):
                acc = None
                for q in range(A.shape[1]):
                    p = f32(A[r, q] * B[q, c])
                    acc = p if acc is None else f32(acc + p)
                out[r, c] = acc
        return Vector._wrap(out[:, 0]) if vec else Matrix._wrap(out)

    def transpose(s): return Matrix._wrap(s.a.T.copy())

    def determinant(s):
        a = s.a
        return f32(f32(f32(a[0, 0] * f32(f32(a[1, 1] * a[2, 2]) - f32(a[2, 1] * a[1, 2])))
                       - f32(a[1, 0] * f32(f32(a[0, 1] * a[2, 2]) - f32(a[2, 1] * a[0, 2])))) +
                   f32(a[2, 0] * f32(f32(a[0, 1] * a[1, 2]) - f32(a[1, 1] * a[0, 2]))))

    def inverse(s):
        a = s.a
        assert a.shape == (3, 3)
        inv_det = f32(1.0) / s.determinant()
        out = _np.zeros((3, 3), f32)
        E = lambda x, y: a[x % 3, y % 3]
        for r in range(3):
            for c in range(3):
                out[c, r] = f32(inv_det * f32(f32(E(r + 1, c + 1) * E(r + 2, c + 2)) - f32(E(r + 2, c + 1) * E(r + 1, c + 2))))
        return Matrix._wrap(out)

    @staticmethod
    def field(n, m, dtype, shape=None):
        return Field(dtype, (n, m), shape)


# -------------------------------------------------------------- scalar / misc ops
def _is_int(x):
    return isinstance(x, (int, _np.integer)) and not isinstance(x, bool)


def sqrt(x):
    if isinstance(x, Tensor):
        return type(x)._wrap(_np.sqrt(x.a))
    with _np.errstate(all="ignore"):
        return f32(_np.sqrt(f32(x)))


cos, sin, tan, acos, exp, log = _m1("cosf"), _m1("sinf"), _m1("tanf"), _m1("acosf"), _m1("expf"), _m1("logf")
_powf, atan2 = _m2("powf"), _m2("atan2f")


def pow(x, y):
    """pow with a constant integer exponent is exponentiation by squaring (what Taichi's simplifier emits), else powf.
    (Scalars reach this through f32.__pow__ as well, so `x ** 2` in the reference is x * x here as it is under Taichi.)"""
    if _is_int(y) and not isinstance(y, bool) and 0 < y <= 16:
        result, base, n = None, x, int(y)
        while n:
            if n & 1:
                result = base if result is None else result * base
            n >>= 1
            if n:
                base = base * base
        return result
    return _powf(x, y)


def floor(x, dtype=None):
    if isinstance(x, Tensor):
        return type(x)._wrap(_np.floor(x.a).astype(f32))
    return f32(_np.floor(f32(x)))


def cast(x, dtype):
    """ti.cast: float -> int truncates toward zero (C semantics), int -> float converts to f32"""
    to_int = dtype in (int, i32)
    if isinstance(x, Tensor):
        return type(x)._wrap(_np.trunc(x.a).astype(_np.int32) if to_int else x.a.astype(f32))
    return int(x) if to_int else f32(x)


def abs(x):
    if isinstance(x, Tensor):
        return type(x)._wrap(_np.abs(x.a))
    return x.__class__(-x if x < 0 else x) if _is_int(x) else f32(_np.abs(f32(x)))


def _mm(fn, pyfn):
    def g(a, b, *rest):
        if rest:
            return g(g(a, b), *rest)
        if isinstance(a, Tensor) or isinstance(b, Tensor):
            cls = type(a) if isinstance(a, Tensor) else type(b)
            return cls._wrap(fn(_raw(a), _raw(b)).astype(f32))
        if _is_int(a) and _is_int(b):
            return pyfn(a, b)
        return f32(fn(f32(a), f32(b)))
    return g


max = _mm(_np.fmax, lambda a, b: a if a > b else b)
min = _mm(_np.fmin, lambda a, b: a if a < b else b)


def select(cond, a, b):
    if isinstance(cond, _np.ndarray):
        cls = type(a) if isinstance(a, Tensor) else (type(b) if isinstance(b, Tensor) else Vector)
        return cls._wrap(_np.where(cond, _raw(a), _raw(b)).astype(f32))
    r = a if cond else b
    if isinstance(r, Tensor):
        return r.copy()
    if isinstance(a, Tensor) or isinstance(b, Tensor):      # scalar broadcast into a vector slot
        t = a if isinstance(a, Tensor) else b
        return type(t)._wrap(_np.full(t.a.shape, f32(r), f32))
    if _is_int(a) and _is_int(b):
        return r
    if isinstance(r, (bool, _np.bool_)):
        return r
    return f32(r)


def sign(x):
    return f32(_np.sign(f32(x)))


def dot(a, b): return (a * b).sum()


def cross(a, b):
    x, y = a.a, b.a
    return Vector._wrap(_np.array([f32(x[1] * y[2]) - f32(x[2] * y[1]), f32(x[2] * y[0]) - f32(x[0] * y[2]),
                                   f32(x[0] * y[1]) - f32(x[1] * y[0])], dtype=f32))


def isnan(x): return _np.isnan(_raw(x))
def isinf(x): return _np.isinf(_raw(x))
def mix(a, b, t): return a * (1 - t) + b * t


# ------------------------------------------------------------------------- RNG
class _Rng:
    def __init__(self):
        self.mode = "script"
        self.script, self.pos = [], 0
        self.key0 = self.key1 = self.ctr0 = 0
        self.draw = 0
        self.log = None

    def set_script(self, values):
        self.mode, self.script, self.pos, self.draw = "script", list(values), 0, 0

    def set_philox(self, pixel, seed, sample):
        self.mode, self.key0, self.key1, self.ctr0, self.draw = "philox", int(pixel), int(seed), int(sample), 0
        self._blk, self._cache = None, None

    @staticmethod
    def philox(c, k):
        M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xffffffff
        c0, c1, c2, c3 = c
        k0, k1 = k
        for _ in range(10):
            p0, p1 = M0 * c0, M1 * c2
            c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
            k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
        return (c0, c1, c2, c3)

    def u32(self):
        d = self.draw
        self.draw += 1
        blk = d >> 2
        if blk != self._blk:
            self._cache, self._blk = self.philox((self.ctr0, blk, 0, 0), (self.key0, self.key1)), blk
        return self._cache[d & 3]

    def next(self, dtype):
        if self.mode == "script":
            v = self.script[self.pos] if self.pos < len(self.script) else (0.5 if dtype is float else 0)
            self.pos += 1
            self.draw += 1
            r = f32(v) if dtype is float else int(v)
        else:
            u = self.u32()
            if dtype is float:
                r = f32(u >> 8) * f32(1.0 / 16777216.0)
            else:
                r = u - (1 << 32) if u >= (1 << 31) else u
        if self.log is not None:
            self.log.append(r)
        return r


RNG = _Rng()
PIXEL_HOOK = [None]        # callable(i, j) invoked when a struct-for over a 2-D field advances


def random(dtype=float):
    return RNG.next(float if dtype in (float, f32) else int)


# ---------------------------------------------------------------- fields / SNodes
class Field:
    def __init__(self, dtype, elem_shape=(), shape=None):
        self.dtype = int if dtype in (int, _np.int32) else f32
        self.elem_shape = tuple(elem_shape)
        self.data = None
        self.snode = None
        if shape is not None:
            self._alloc(shape if isinstance(shape, (tuple, list)) else (shape,))

    def _alloc(self, shape):
        self.shape = tuple(shape)
        self.data = _np.zeros(self.shape + self.elem_shape, dtype=_np.int32 if self.dtype is int else f32)
        self.active = _np.zeros(self.shape if self.shape else (1,), dtype=bool)

    def _key(self, idx):
        if idx is None:
            return ()
        return idx if isinstance(idx, tuple) else (idx,)

    def __getitem__(self, idx):
        v = self.data[tuple(int(q) for q in self._key(idx))]
        if self.elem_shape == ():
            return int(v) if self.dtype is int else f32(v)
        return (Vector if len(self.elem_shape) == 1 else Matrix)._wrap(v.copy())

    def __setitem__(self, idx, val):
        k = tuple(int(q) for q in self._key(idx))
        self.data[k] = _raw(val)
        if self.shape:
            self.active[k] = True

    def from_numpy(self, arr): self.data[...] = arr
    def to_numpy(self): return self.data.copy()
    def fill(self, v): self.data[...] = v

    def __iter__(self):
        hook = PIXEL_HOOK[0]
        for idx in _np.ndindex(*self.shape):
            if hook is not None and len(idx) == 2:
                hook(*idx)
            yield idx if len(idx) > 1 else idx[0]


def field(dtype, shape=None):
    return Field(dtype, (), shape)


class StructField:
    def __init__(self, cls):
        self.cls, self.items, self.shape, self.snode = cls, None, None, None

    def _alloc(self, shape):
        assert len(shape) == 1
        self.shape = tuple(shape)
        self.items = [self.cls() for _ in range(shape[0])]
        self.active = _np.zeros(shape, dtype=bool)

    def __getitem__(self, idx): return self.items[int(idx)]

    def __setitem__(self, idx, val):
        self.items[int(idx)] = val
        self.active[int(idx)] = True

    def __iter__(self): return iter(range(self.shape[0]))


class SNode:
    def __init__(self, shape=()):
        self.shape, self.placed = tuple(shape), []

    def _child(self, axes, dims):
        axes = axes if isinstance(axes, (tuple, list)) else (axes,)
        dims = dims if isinstance(dims, (tuple, list)) else (dims,) * len(axes)
        return SNode(self.shape + tuple(int(d) for d in dims))

    dense = bitmasked = pointer = _child

    def place(self, *fields):
        for f in fields:
            f._alloc(self.shape)
            f.snode = self
            self.placed.append(f)
        return self


root = SNode()


def is_active(snode, idx):
    k = idx if isinstance(idx, (tuple, list)) else (idx,)
    return any(bool(f.active[tuple(int(q) for q in k)]) for f in snode.placed)


# ------------------------------------------------------------------ decorators
def _as_f32(v):
    """Inside a kernel every float is an f32 value: Python floats leaving a @ti.func become np.float32 (a Python
    float would also raise on x / 0.0 where Taichi, like numpy, yields inf / nan)."""
    if isinstance(v, float):
        return f32(v)
    if isinstance(v, tuple):
        return tuple(_as_f32(x) for x in v)
    return v


def func(f):
    import functools

    @functools.wraps(f)
    def wrapped(*a, **k):
        return _as_f32(f(*a, **k))
    return wrapped


def kernel(f): return f
def data_oriented(c): return c
pyfunc = func
def static(x, *rest): return x if not rest else (x,) + rest
def template(): return None
def loop_config(**kw): pass
def init(**kw): pass
def static_assert(*a): pass


class _Experimental:
    @staticmethod
    def real_func(f): return f


experimental = _Experimental()


def _default_for(ann):
    if ann is int:
        return 0
    if ann in (float, f32):
        return f32(0)
    if isinstance(ann, _TensorType):
        return ann.zero()
    if isinstance(ann, type) and getattr(ann, "_is_ti_struct", False):
        return ann()
    return 0


def _coerce(ann, v):
    if ann is int:
        return int(v)
    if ann in (float, f32):
        return f32(v)
    if isinstance(ann, _TensorType):
        return ann(v)
    if isinstance(v, Tensor):
        return v.copy()
    return v


def dataclass(cls):
    """@ti.dataclass: typed members with zero defaults, copy-on-read for tensors."""
    ann = dict(getattr(cls, "__annotations__", {}))

    def __init__(self, **kw):
        for name, a in ann.items():
            object.__setattr__(self, name, _coerce(a, kw[name]) if name in kw else _default_for(a))
        for name in kw:
            if name not in ann:
                raise TypeError(f"{cls.__name__}: unknown member {name}")

    def __getattribute__(self, name):
        v = object.__getattribute__(self, name)
        return v.copy() if isinstance(v, Tensor) else v

    def __setattr__(self, name, v):
        a = ann.get(name)
        object.__setattr__(self, name, _coerce(a, v) if a is not None else v)

    cls.__init__, cls.__getattribute__, cls.__setattr__ = __init__, __getattribute__, __setattr__
    cls._is_ti_struct = True
    cls.field = classmethod(lambda c, shape=None: StructField(c))
    return cls


# ------------------------------------------------------------------------ types
class _TensorType:
    def __init__(self, shape, dtype=float):
        self.shape, self.dtype = tuple(shape), dtype

    def zero(self):
        if len(self.shape) == 1:
            return Vector._wrap(_np.zeros(self.shape, f32))
        return Matrix._wrap(_np.zeros(self.shape, f32))

    def __call__(self, *args):
        if len(args) == 1:
            d = args[0]
            arr = _np.array(_raw(d) if not isinstance(d, (list, tuple)) else [[_raw(x) for x in r] if isinstance(r, (list, tuple)) else _raw(r) for r in d], dtype=f32)
            if arr.ndim == 0:
                arr = _np.full(self.shape, arr, f32)
        else:
            arr = _np.array([_raw(x) for x in args], dtype=f32)
        arr = arr.reshape(self.shape).copy()
        return Vector._wrap(arr) if len(self.shape) == 1 else Matrix._wrap(arr)


class _Types:
    @staticmethod
    def vector(n, dtype=float): return _TensorType((n,), dtype)
    @staticmethod
    def matrix(n, m, dtype=float): return _TensorType((n, m), dtype)
    @staticmethod
    def ndarray(**kw): return None


types = _Types()
from . import math  # noqa: E402,F401
