// vec.hpp — float3 algebra for the gfx950 path tracer.
//
// Operation ORDER is part of the contract: sums/dots accumulate left to right and
// normalize() multiplies by a reciprocal length, which is what the reference's Taichi
// vector ops do (see DESIGN.md "float parity").  Compiled with -ffp-contract=off, so what is
// written here is what the VALU executes; do not "simplify" expressions.
#pragma once
#include <hip/hip_runtime.h>

// APT_FAST (set by adapt_amd/build.py): 1 = the product build - small-scene intersectors re-derived for speed inside SURVEY 8(d)'s
// tolerances (traverse.hpp "Flat sweep"); 0 = the exact build, whose intersectors are the reference's loop operation for operation (the
// build the bit-exact parity tests pin).  The shading arithmetic is the same in both except for the transcendental calls and the sdiv / ssqrt helpers below (product build: float functions, 1-ulp division and roots outside delta interactions).  See DESIGN.md "float parity policy".
#ifndef APT_FAST
#define APT_FAST 0
#endif

#define APT_HD __host__ __device__ __forceinline__
#define APT_D __device__ __forceinline__

struct f3 {
    float x, y, z;
};

// Divisions and square roots of the non-delta SHADING code (light sampling, MIS weights, lobe sampling and evaluation, throughput, roulette).
// APT_FAST_DIV=1 (device code): a * v_rcp_f32(b), v_sqrt_f32, v_rsq_f32 - 1 ulp each, the same infinities, zeros and NaNs as the IEEE forms for
// every operand but denormal divisors - where the IEEE sequences are ~10 instructions with a dependent chain (a vertex of the Cornell box makes
// ~15 divisions and 5 roots).  The intersectors never come through here: their reference-order code divides with `/` and sqrtf() in both builds.
#ifndef APT_FAST_DIV
#define APT_FAST_DIV 0
#endif
#if APT_FAST_DIV && defined(__HIP_DEVICE_COMPILE__)
APT_HD float sdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
APT_HD float srcp(float b) { return __builtin_amdgcn_rcpf(b); }
APT_HD float ssqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
APT_HD float srsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#else
APT_HD float sdiv(float a, float b) { return a / b; }
APT_HD float srcp(float b) { return 1.0f / b; }
APT_HD float ssqrt(float x) { return sqrtf(x); }
APT_HD float srsqrt(float x) { return 1.0f / sqrtf(x); }
#endif

APT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
APT_HD f3 splat3(float s) { return mk3(s, s, s); }
APT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
APT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
APT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
APT_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
APT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
APT_HD f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
APT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
APT_HD f3 operator+(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
APT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
APT_HD float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
APT_HD float norm2(f3 a) { return dot(a, a); }
APT_HD float norm(f3 a) { return sqrtf(norm2(a)); }
APT_HD f3 normalize(f3 a) { float inv = 1.0f / norm(a); return a * inv; }
// the shading code's flavours (sdiv / ssqrt above): light sampling, MIS, lobe sampling, throughput.  NOT used by delta interactions (mirror,
// glass: shading.hpp keeps `/`, sqrtf and normalize() there) - a perturbation at EVERY vertex of a specular chain compounds (measured: C3 every
// pixel 99.7 % -> 97.8 % with them fast as well), one that enters a chain from a diffuse vertex does not (the float sin / cos of the lobe samples)
#if APT_FAST_DIV && defined(__HIP_DEVICE_COMPILE__)
APT_HD f3 fdiv3(f3 a, float s) { const float r = srcp(s); return a * r; }
#else
APT_HD f3 fdiv3(f3 a, float s) { return a / s; }                 // (the reference divides every component: three IEEE divisions, not one reciprocal)
#endif
APT_HD float fnorm(f3 a) { return ssqrt(norm2(a)); }
APT_HD f3 fnormalize(f3 a) { return a * srsqrt(norm2(a)); }
APT_HD f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
APT_HD float max3(f3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
APT_HD float min3(f3 a) { return fminf(fminf(a.x, a.y), a.z); }
APT_HD f3 abs3(f3 a) { return mk3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
APT_HD f3 min3v(f3 a, f3 b) { return mk3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
APT_HD f3 max3v(f3 a, f3 b) { return mk3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
APT_HD float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
APT_HD float sqr(float x) { return x * x; }

struct m33 {
    float m[3][3];
};
APT_HD f3 mul(const m33& M, f3 a) {
    return mk3((M.m[0][0] * a.x + M.m[0][1] * a.y) + M.m[0][2] * a.z,
               (M.m[1][0] * a.x + M.m[1][1] * a.y) + M.m[1][2] * a.z,
               (M.m[2][0] * a.x + M.m[2][1] * a.y) + M.m[2][2] * a.z);
}

// ---- transcendental functions.
// APT_EXACT_MATH=1 (the exact build): evaluate in double and round once.  That reproduces what a correctly rounded float libm returns, so
// the HIP path and the CPU oracle (glibc) agree bit-for-bit on essentially every call; the cost is a handful of FP64 ops per bounce.
// APT_EXACT_MATH=0 (the product build since round 6): OCML's native float versions (1-2 ulp): parity statistical, and measured to be the
// double version's (adapt_amd/build.py); the reference itself runs Taichi's fast-math float functions.
#ifndef APT_EXACT_MATH
#define APT_EXACT_MATH 1
#endif
#if APT_EXACT_MATH
APT_D float apt_cos(float x) { return (float)cos((double)x); }
APT_D float apt_sin(float x) { return (float)sin((double)x); }
// sin and cos of an azimuth in [0, 2 pi] (the only use: polar_dir), in double, rounded once.  Same contract as the generic
// versions above, but OCML's double sincos carries its large-argument reduction through the whole shade kernel (+16 VGPRs,
// one occupancy step); for |x| < 8 two fused steps against a two-word pi/2 and the fdlibm kernels on [-pi/4, pi/4]
// (error < 1 ulp in double, i.e. invisible after the rounding to float except on ~1e-8 of arguments) are enough.
APT_D void apt_sincos(float xf, float* s, float* c) {
#ifdef APT_SINCOS_OCML
    { double ds, dc; sincos((double)xf, &ds, &dc); *s = (float)ds; *c = (float)dc; return; }
#endif
    const double x = (double)xf;
    const double k = __builtin_rint(x * 6.36619772367581382433e-01);                 // nearest multiple of pi/2
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06); ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03); ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(z * r, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07); pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03); pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cs = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)k & 3;
    const double so = (q & 1) ? cs : sn, co = (q & 1) ? sn : cs;
    *s = (float)((q & 2) ? -so : so);
    *c = (float)(((q + 1) & 2) ? -co : co);
}
APT_D float apt_tan(float x) { return (float)tan((double)x); }
APT_D float apt_pow(float x, float y) { return (float)pow((double)x, (double)y); }
#else
APT_D float apt_cos(float x) { return cosf(x); }
APT_D float apt_sin(float x) { return sinf(x); }
APT_D void apt_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
APT_D float apt_tan(float x) { return tanf(x); }
APT_D float apt_pow(float x, float y) { return powf(x, y); }
#endif
// b ^ (e.x, e.y, e.z).  Glossiness exponents are almost always one number broadcast to rgb (`<rgb name="k_g" value="10.0"/>`): one
// pow serves the three channels then (the double-precision pow is the most expensive call of the Phong models: the Blinn-Phong class
// kernel of BASELINE C3 issued 44 k VALU instructions per wave with three calls per evaluation), and x^1 is x itself, exactly, in
// any correctly-rounding pow.
APT_D f3 pow_sv(float b, f3 e) {
    if (e.x == e.y && e.y == e.z) return splat3((e.x == 1.0f) ? b : apt_pow(b, e.x));
    return mk3(apt_pow(b, e.x), apt_pow(b, e.y), apt_pow(b, e.z));
}

#define APT_PI      ((float)3.14159265358979323846)
#define APT_INV_PI  ((float)(1.0 / 3.14159265358979323846))
#define APT_INV_2PI ((float)((1.0 / 3.14159265358979323846) * 0.5))
#define APT_2PI     ((float)(2.0 * 3.14159265358979323846))
#define APT_PI_2    ((float)(3.14159265358979323846 / 2.0))
