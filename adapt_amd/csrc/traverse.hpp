// traverse.hpp — software ray traversal for gfx950 (no RT hardware on CDNA4).
//
// Replaces the reference's intersectors: brute force `TracerBase.ray_intersect/does_intersect`
// (tracer/tracer_base.py:168-278) and the stackless preorder walk
// `PathTracer.ray_intersect_bvh/does_intersect_bvh` (tracer/path_tracer.py:338-422).
// The per-primitive tests are the reference's, operation for operation (triangle: solve
// [e1 e2 -d] (u v t)^T = o - p0 with the adjugate inverse; sphere: tracer_base.py:184-199),
// so the closest hit is the same hit; the tree around them is ours (bvh_build.cpp -> bvh_wide.cpp):
//   * an 8-wide tree with child boxes quantised to 8 bits per plane: one 64-byte node fetch (four 16-byte pieces, layout in
//     bvh_wide.cpp) decides up to eight subtrees.  A closest-hit ray of the 95 k-triangle scene makes ~5 node steps where the binary
//     tree made ~15; what a step costs the CU is its scattered 16-byte accesses (one lane-access per cycle through the vector-memory
//     pipe, whatever the bytes: tools/probe_gather.hip), hence the node's size;
//   * node corners live on a global 16-bit grid and the ray is moved into grid units once (make_walk_ray), so a plane of any node is
//     the exact small integer `corner + q * 2^e` and the slab test is one FMA per plane: t = q * (2^e / d') + (corner - o') / d';
//   * front-to-back order without sorting: child slots are assigned by octant at build time, a ray visits the hit slots in the
//     order `slot XOR ray octant`; the hits of a node travel as one bit mask ("node group"), its hit leaf primitives as another
//     ("triangle group"), and the per-lane stack holds 8-byte groups, in LDS laid out [level][lane] (bank-conflict free), deep
//     levels spilling to a per-lane global column;
//   * box tests are conservative, not exact: decoded boxes contain the builder's boxes, which are padded by 1e-4 + 1e-5 |x| —
//     three orders of magnitude more than the rounding of the fused slab arithmetic — so no box that contains a hit is ever
//     culled and the RESULT is that of the exact per-primitive tests alone.
#pragma once
#include "vec.hpp"

struct DevBvh {
    const uint4* nodes;      // 4 uint4 (64 bytes) per node, layout in bvh_wide.cpp
    const float4* prims;     // 3 float4 per primitive, leaf order
    const int* slot_prim;    // leaf-order slot -> primitive index (product build: its records carry no id, the walk reports the slot and the winner is looked up once per ray)
    int n_nodes, n_prims;
    float gmin[3], ginv[3], gstep[3];      // the global grid of the node corners: grid = (world - gmin) * ginv, gstep = 1 / ginv (powers of two)
};
// primitive record, exact build:   triangle q0=(p0, e1.x) q1=(e1.yz, e2.xy) q2=(e2.z, prim_id, 0, -)
//                                  sphere   q0=(centre, r)                  q2=(-, prim_id, 1, -)
// primitive record, product build: triangle q0=(p0, U.x)  q1=(U.yz, V.xy)   q2=(V.z, T.xyz)      rows U, V, T of [e1 e2 n]^-1 (flat_build.cpp planar_rows)
//                                  sphere   q0=(centre, NaN) q1=(r, -, -, -)
#ifndef APT_FAST_LEAVES
#define APT_FAST_LEAVES APT_FAST     // product build: precomputed-transform leaf records (0: the exact build's records and test in the product build, measurement only)
#endif
struct HitRec { float t; int prim; float u, v; };

typedef float v2f __attribute__((ext_vector_type(2)));
APT_D v2f sp2(float s) { v2f r; r.x = s; r.y = s; return r; }
APT_D v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }

// Ray as the walk wants it.  The reciprocal direction is clamped to +-1e30 so that axis-parallel rays produce huge but finite slab
// distances of the right sign (inf would turn the fused form q * s + c into inf - inf = NaN and the axis would stop culling).
struct WalkRay { f3 o, d, inv, noo; uint32_t octinv4; };
#ifdef APT_WALK_IEEE_RCP
APT_D float walk_rcp(float d) { return (fabsf(d) < 1e-30f) ? copysignf(1e30f, d) : 1.0f / d; }
#else
// v_rcp_f32 (1 ulp) is enough: the reciprocal only feeds the conservative box tests, whose margin (the builder's 1e-4 padding) is
// three orders of magnitude above it; the exact per-primitive tests divide for themselves
APT_D float walk_rcp(float d) { return (fabsf(d) < 1e-30f) ? copysignf(1e30f, d) : __builtin_amdgcn_rcpf(d); }
#endif
APT_D WalkRay make_walk_ray(const DevBvh& b, f3 o, f3 d) {
    WalkRay r; r.o = o; r.d = d;
    // inv and noo are the ray in GRID units (a plane X of the tree is world gmin + gstep * X): t = (X - o') * (gstep / d), o' = (o - gmin) / gstep
    r.inv = mk3(walk_rcp(d.x) * b.gstep[0], walk_rcp(d.y) * b.gstep[1], walk_rcp(d.z) * b.gstep[2]);
    const f3 og = mk3((o.x - b.gmin[0]) * b.ginv[0], (o.y - b.gmin[1]) * b.ginv[1], (o.z - b.gmin[2]) * b.ginv[2]);
    r.noo = mk3(-(og.x * r.inv.x), -(og.y * r.inv.y), -(og.z * r.inv.z));
    // a ray with a NaN or infinite component hits nothing (as under the compare-based test of rounds 2-5, where NaN slabs failed every `tn <= tf`;
    // the sign-bit test below would let v_max / v_min drop the NaNs and walk the whole tree): every entry distance huge
    if (!(fabsf((r.noo.x + r.noo.y + r.noo.z) + (r.inv.x + r.inv.y + r.inv.z)) < 3e38f)) { r.inv = splat3(0.f); r.noo = splat3(3e38f); }
    const uint32_t oct = (r.inv.x < 0.f ? 4u : 0u) | (r.inv.y < 0.f ? 2u : 0u) | (r.inv.z < 0.f ? 1u : 0u);
    r.octinv4 = (7u - oct) * 0x01010101u;
    return r;
}

// One primitive against the ray.  Returns the reference's ray_t (or -1) and barycentrics.
// sphere: tracer_base.py:184-199 (the reference's arithmetic in both builds: a grazing hit is a difference of two nearly equal squares)
APT_D float sphere_test_t(f3 c, float radius, f3 o, f3 d) {
    float r2 = radius * radius;
    f3 s2c = c - o;
    float cn2 = norm2(s2c);
    float proj = dot(d, s2c);
    float c2ray = cn2 - proj * proj;
    if (c2ray >= r2) return -1.f;
    float cut = sqrtf(r2 - c2ray);
    return proj + ((cn2 > r2 + 1e-4f) ? -cut : cut);
}
APT_D float prim_test(float4 q0, float4 q1, float4 q2, f3 o, f3 d, float& u, float& v) {
    u = 0.f; v = 0.f;
    if (__float_as_int(q2.z) != 0) return sphere_test_t(mk3(q0.x, q0.y, q0.z), q0.w, o, d);
    // triangle: columns (e1, e2, -d); inverse = adjugate * (1/det), Taichi's 3x3 formula
    float a00 = q0.w, a10 = q1.x, a20 = q1.y;      // e1
    float a01 = q1.z, a11 = q1.w, a21 = q2.x;      // e2
    float a02 = -d.x, a12 = -d.y, a22 = -d.z;
    float c00 = a11 * a22 - a21 * a12, c01 = a21 * a02 - a01 * a22, c02 = a01 * a12 - a11 * a02;
    float det = (a00 * c00 + a10 * c01) + a20 * c02;
    float inv_det = 1.0f / det;
    f3 s = o - mk3(q0.x, q0.y, q0.z);
    float c10 = a12 * a20 - a22 * a10, c11 = a22 * a00 - a02 * a20, c12 = a02 * a10 - a12 * a00;
    float c20 = a10 * a21 - a20 * a11, c21 = a20 * a01 - a00 * a21, c22 = a00 * a11 - a10 * a01;
    u = ((inv_det * c00) * s.x + (inv_det * c01) * s.y) + (inv_det * c02) * s.z;
    v = ((inv_det * c10) * s.x + (inv_det * c11) * s.y) + (inv_det * c12) * s.z;
    float t = ((inv_det * c20) * s.x + (inv_det * c21) * s.y) + (inv_det * c22) * s.z;
    return (u >= 0.f && v >= 0.f && u + v <= 1.0f) ? t : -1.f;
}

// Per-lane traversal stack of 8-byte groups: the first `k` levels live in LDS ([level][lane]), deeper levels - rare - spill to a
// per-lane column of a global buffer.  The two halves are addressed through typed pointers (ds_read_b64 / ds_write_b64 and global
// accesses; a generic pointer would make every push and pop a flat access plus an address select).
typedef unsigned int grp_t __attribute__((ext_vector_type(2)));      // (x, y) group word pair: a plain vector type, so that it can live behind address-space pointers
APT_D grp_t mk_grp(uint32_t x, uint32_t y) { grp_t g; g.x = x; g.y = y; return g; }
typedef __attribute__((address_space(3))) grp_t lds_u2;
typedef __attribute__((address_space(1))) grp_t glb_u2;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
// lut: 8 x 256 bytes in LDS, lut[oi * 256 + x] = the bits of x moved from position s to s ^ oi (node8_test's priority order), or null
struct TravStack { lds_u2* lds; int stride; glb_u2* ovf; int ovf_stride; int k; lds_u8* lut; };
APT_D uint32_t xor_permute8(uint32_t x, uint32_t oi) {            // bit `slot` -> bit `slot ^ oi`: three conditional swaps of bit groups
    x = (oi & 4u) ? (((x << 4) | (x >> 4)) & 0xffu) : x;
    x = (oi & 2u) ? (((x & 0x33u) << 2) | ((x >> 2) & 0x33u)) : x;
    x = (oi & 1u) ? (((x & 0x55u) << 1) | ((x >> 1) & 0x55u)) : x;
    return x;
}
// every thread of a 256-thread workgroup fills its column of the table (callers synchronise before the first walk)
APT_D void fill_permute_lut(lds_u8* lut) { for (uint32_t oi = 0; oi < 8u; oi++) lut[oi * 256u + threadIdx.x] = (unsigned char)xor_permute8(threadIdx.x & 0xffu, oi); }
APT_D void tpush(const TravStack& s, int& sp, grp_t v) {
    if (sp < s.k) s.lds[sp * s.stride] = v; else s.ovf[(sp - s.k) * s.ovf_stride] = v;
    sp++;
}
APT_D grp_t tpop(const TravStack& s, int& sp) {
    sp--;
    grp_t v;
    if (sp < s.k) v = s.lds[sp * s.stride]; else v = s.ovf[(sp - s.k) * s.ovf_stride];
    return v;
}
#ifdef APT_WALK_STATS
#define WALK_COUNT(x) ((x)++)
#else
#define WALK_COUNT(x) ((void)0)
#endif
struct WalkStats { uint32_t nodes, prims; };

// record strides as shift-adds: v_mul_lo_u32 issues at quarter rate and sits at the head of every dependent fetch of the walk
APT_D uint32_t mul48(uint32_t i) { return (i << 5) + (i << 4); }
APT_D float ubyte_f(uint32_t w, int b) { return (float)((w >> (8 * b)) & 0xffu); }       // v_cvt_f32_ubyte<b>

// Fetch node `idx` and test its eight child boxes against the ray segment [0, tmax].
// Out: ng = node group of the node's INNER children (x = index of the first one; y = the hit ones in bits 31..24, at position
// 24 + (slot ^ (7 - ray octant)) so that "highest bit first" is front to back, and the node's inner-slot mask in bits 7..0);
// tg = triangle group of its LEAF children (x = index of the first one's primitive | the node's leaf-slot mask << 24; y = the hit ones,
// bit = slot).  Child k of either kind is `first + popcount(mask & ((1 << slot) - 1))`.
APT_D void node8_test(const DevBvh& b, const TravStack& ts, uint32_t idx, const WalkRay& r, float tmax, grp_t& ng, grp_t& tg) {
    const char* base = reinterpret_cast<const char*>(b.nodes) + (idx << 6);          // wave-uniform base + 32-bit offset
    const uint4 n0 = *reinterpret_cast<const uint4*>(base), n1 = *reinterpret_cast<const uint4*>(base + 16), n2 = *reinterpret_cast<const uint4*>(base + 32),
                n3 = *reinterpret_cast<const uint4*>(base + 48);
    const uint32_t lmask = (n0.y >> 16) & 0xffu, imask = n0.y >> 24;
    // slab arithmetic in the node's frame (grid units): t = q * (2^e * inv) + (corner * inv + noo)
    const float sx = ldexpf(r.inv.x, (int)((n0.z >> 24) & 15u)), sy = ldexpf(r.inv.y, (int)(n0.z >> 28)), sz = ldexpf(r.inv.z, (int)((n0.w >> 24) & 15u));
    const float cx = __builtin_fmaf((float)(n0.x & 0xffffu), r.inv.x, r.noo.x), cy = __builtin_fmaf((float)(n0.x >> 16), r.inv.y, r.noo.y),
                cz = __builtin_fmaf((float)(n0.y & 0xffffu), r.inv.z, r.noo.z);
    // near / far planes by the sign of the direction, four children per dword
    const bool nx = r.inv.x < 0.f, ny = r.inv.y < 0.f, nz = r.inv.z < 0.f;
    const uint32_t nearx[2] = {nx ? n2.z : n1.x, nx ? n2.w : n1.y}, farx[2] = {nx ? n1.x : n2.z, nx ? n1.y : n2.w};
    const uint32_t neary[2] = {ny ? n3.x : n1.z, ny ? n3.y : n1.w}, fary[2] = {ny ? n1.z : n3.x, ny ? n1.w : n3.y};
    const uint32_t nearz[2] = {nz ? n3.z : n2.x, nz ? n3.w : n2.y}, farz[2] = {nz ? n2.x : n3.z, nz ? n2.y : n3.w};
    // one bit per child, child 7 first: the sign of (exit - entry) is shifted in from the right, so that child c ends up in bit c
    // (two instructions per child where compare + select + shift + or were six)
    uint32_t miss = 0;
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        const int h = k >> 2, c = k & 3;
        const v2f tx = __builtin_elementwise_fma(mk2(ubyte_f(nearx[h], c), ubyte_f(farx[h], c)), sp2(sx), sp2(cx));
        const v2f ty = __builtin_elementwise_fma(mk2(ubyte_f(neary[h], c), ubyte_f(fary[h], c)), sp2(sy), sp2(cy));
        const v2f tz = __builtin_elementwise_fma(mk2(ubyte_f(nearz[h], c), ubyte_f(farz[h], c)), sp2(sz), sp2(cz));
        const float tn = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.f);
        const float tf = fminf(fminf(fminf(tx.y, ty.y), tz.y), tmax);
        miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf - tn), 31u);
    }
    const uint32_t hit_l = ~miss & lmask;
    uint32_t x = ~miss & imask;
    // inner hits to their priority positions: bit `slot` -> bit `slot ^ (7 - octant)` - one byte from a 2 KiB table in LDS where the kernel keeps one
    // (the walk kernels: 3 instructions for 15; the walk is bound by the issue of its node steps), three conditional bit-group swaps elsewhere
    if (ts.lut) x = ts.lut[(r.octinv4 & 0x700u) | x];
    else x = xor_permute8(x, r.octinv4 & 7u);
    ng.x = n0.z & 0x00ffffffu; ng.y = (x << 24) | imask;
    tg.x = (n0.w & 0x00ffffffu) | (lmask << 24); tg.y = hit_l;
}
// the primitive a triangle group's bit `slot` stands for (leaf order), and the bit taken out of the group
APT_D uint32_t tri_take(grp_t& tg) {
    const uint32_t k = 31u - (uint32_t)__clz((int)tg.y);
    tg.y &= ~(1u << k);
    return (tg.x & 0x00ffffffu) + (uint32_t)__popc((tg.x >> 24) & ((1u << k) - 1u));
}

#if APT_FAST
APT_D v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#endif
#if APT_FAST_LEAVES
// Product build: the leaf test of the flat sweep (planar_solve below) on the tree's primitives, one pending primitive of a lane per call
// (tri_group below).  Record = corner p0 and the rows U, V, T of [e1 e2 n]^-1 (Baldwin & Weber, JCGT 2016; computed in
// double on the host): s = o - p0 (the reference's own first operation, tracer_base.py:206: a ray that STARTS on the primitive keeps its
// height T . s a sum of small products), t = -T.s / T.d with ONE reciprocal, then (u, v) = (U.P, V.P) at the hit point P = s + t d -
// ~40 instructions per pair where the adjugate solve with its IEEE division takes ~95.  Inside SURVEY 8(d): t within 1e-5 relative of the
// exact build's, same primitive unless tied (tests/test_gpu_fast.py).  Of two primitives at the same distance to the last bit - a ray
// through a shared edge that both inside tests accept: a strip about 1e-7 of the edge length wide - the first one tested wins (the exact
// build reproduces upstream's "lower index"; under different arithmetic the two distances are no longer equal to the bit anyway).
// Records carry no primitive id: rec.prim is the LEAF SLOT while a ray walks, the callers translate the winner (DevBvh::slot_prim).
APT_D float walk_scalar_test(float4 q0, float4 q1, float4 q2, f3 o, f3 d, float& u, float& v) {      // one record, either kind (lanes that meet a sphere)
    u = 0.f; v = 0.f;
    if (q0.w != q0.w) return sphere_test_t(mk3(q0.x, q0.y, q0.z), q1.x, o, d);
    const f3 s = o - mk3(q0.x, q0.y, q0.z);
    const float t_o = __builtin_fmaf(q2.y, s.x, __builtin_fmaf(q2.z, s.y, q2.w * s.z));
    const float t_d = __builtin_fmaf(q2.y, d.x, __builtin_fmaf(q2.z, d.y, q2.w * d.z));
    const float t = -t_o * __builtin_amdgcn_rcpf(t_d);
    const f3 P = mk3(__builtin_fmaf(t, d.x, s.x), __builtin_fmaf(t, d.y, s.y), __builtin_fmaf(t, d.z, s.z));
    u = __builtin_fmaf(q0.w, P.x, __builtin_fmaf(q1.x, P.y, q1.y * P.z));
    v = __builtin_fmaf(q1.z, P.x, __builtin_fmaf(q1.w, P.y, q2.x * P.z));
    return (fminf(fminf(u, v), (1.0f - u) - v) >= 0.f) ? t : -1.f;
}
template <bool ANY>
APT_D bool tri_one(const DevBvh& b, grp_t& tg, const WalkRay& r, HitRec& rec, WalkStats& ws) {
    const uint32_t slot = tri_take(tg);
    const char* base = reinterpret_cast<const char*>(b.prims) + mul48(slot);
    const float4 p0 = *reinterpret_cast<const float4*>(base), p1 = *reinterpret_cast<const float4*>(base + 16), p2 = *reinterpret_cast<const float4*>(base + 32);
    float u, v;
    const float t = walk_scalar_test(p0, p1, p2, r.o, r.d, u, v);
    WALK_COUNT(ws.prims);
    if (ANY) return t > 1e-4f && t < rec.t;
    if (t > 1e-4f && t < rec.t) { rec.t = t; rec.prim = (int)slot; rec.u = u; rec.v = v; }
    return false;
}
// leaf slot -> primitive index (low 28 bits) and material class (bits 28..30) of a finished closest-hit walk: ONE lookup at the hand-in,
// where the exact build makes one too (prim_class[prim]) - a second dependent load there stalls the whole wave
APT_D int walk_info(const DevBvh& b, int slot) { return slot >= 0 ? b.slot_prim[slot] : -1; }
APT_D int walk_prim(const DevBvh& b, int slot) { return slot >= 0 ? (b.slot_prim[slot] & 0x0fffffff) : -1; }
#else
// One primitive of a triangle group (the highest pending bit).  Closest hit: strictly nearer wins, and of two primitives at
// EXACTLY the same t the lower original index - the reference's brute-force loop keeps the first strictly-closer hit in index
// order (tracer_base.py:208), so the answer does not depend on the order in which the tree presents the primitives.
// Any hit: true at a primitive with 1e-4 < t < rec.t.
template <bool ANY>
APT_D bool tri_one(const DevBvh& b, grp_t& tg, const WalkRay& r, HitRec& rec, WalkStats& ws) {
    const char* base = reinterpret_cast<const char*>(b.prims) + mul48(tri_take(tg));
    const float4 p0 = *reinterpret_cast<const float4*>(base), p1 = *reinterpret_cast<const float4*>(base + 16), p2 = *reinterpret_cast<const float4*>(base + 32);
    float u, v;
    const float t = prim_test(p0, p1, p2, r.o, r.d, u, v);
    WALK_COUNT(ws.prims);
    if (ANY) return t > 1e-4f && t < rec.t;
    const int kid = __float_as_int(p2.y);
    if (t > 1e-4f && (t < rec.t || (t == rec.t && kid < rec.prim))) { rec.t = t; rec.prim = kid; rec.u = u; rec.v = v; }
    return false;
}
APT_D int walk_prim(const DevBvh&, int prim) { return prim; }      // exact build: the records carry the primitive index
#endif
// (One primitive per call.  Rounds 2-3 tested two of a lane's pending primitives in the halves of packed instructions; with the product
// build's cheap leaf test the walk answers to its scattered 16-byte loads, not to arithmetic, and the pair test fetched six whether or
// not the lane had a second primitive pending: C4 extend 22.2 -> 21.1 ms per 64 spp, C5 14.05 -> 13.7 per 32 for the single test, round 4.)
template <bool ANY>
APT_D bool tri_group(const DevBvh& b, grp_t& tg, const WalkRay& r, HitRec& rec, WalkStats& ws) {
    while (tg.y != 0u) if (tri_one<ANY>(b, tg, r, rec, ws)) return true;
    return false;
}

// One traversal step of a lane that holds node group `ng` (x = index of the parent's first inner child, y = hit bits 31..24 |
// the parent's inner-child mask 7..0): take the nearest pending child, leave the rest of the group on the stack, test the child's
// boxes.  A group without inner bits is a postponed triangle group.
#define APT_GROUP_HAS_NODES(g) ((g).y > 0x00ffffffu)
APT_D void group_step(const DevBvh& b, const TravStack& ts, int& sp, grp_t& ng, grp_t& tg, const WalkRay& r, float tmax, WalkStats& ws) {
    if (APT_GROUP_HAS_NODES(ng)) {
        const uint32_t bit = 31u - (uint32_t)__clz((int)ng.y);
        const uint32_t pim = ng.y & 0xffu;
        ng.y &= ~(1u << bit);
        if (APT_GROUP_HAS_NODES(ng)) tpush(ts, sp, ng);
        const uint32_t slot = (bit - 24u) ^ (r.octinv4 & 7u);
        const uint32_t idx = ng.x + (uint32_t)__popc(pim & ((1u << slot) - 1u));
        node8_test(b, ts, idx, r, tmax, ng, tg);
        WALK_COUNT(ws.nodes);
    } else { tg = ng; ng.x = 0u; ng.y = 0u; }
}
#define APT_ROOT_GROUP mk_grp(0u, 0x80000000u)           // "child in priority position 7 of a parent with no inner-child mask" = node 0

// ANY = false: closest hit, rec.t starts at the search limit and ends at min_depth.
// ANY = true : returns true on the first hit with 1e-4 < t < rec.t.
template <bool ANY>
APT_D bool traverse(const DevBvh& bvh, const TravStack& ts, f3 o, f3 d, HitRec& rec) {
    const WalkRay r = make_walk_ray(bvh, o, d);
    WalkStats ws; ws.nodes = ws.prims = 0;
    int sp = 0;
    grp_t ng = APT_ROOT_GROUP, tg = mk_grp(0u, 0u);
    for (;;) {
        group_step(bvh, ts, sp, ng, tg, r, rec.t, ws);
        if (tri_group<ANY>(bvh, tg, r, rec, ws)) return true;
        if (!APT_GROUP_HAS_NODES(ng)) {
            if (sp == 0) break;
            ng = tpop(ts, sp);
        }
    }
    if (!ANY) rec.prim = walk_prim(bvh, rec.prim);
    return false;
}

// ---------------------------------------------------------------------------------------------
// Uniform sweep for small scenes (a few dozen primitives: the Cornell configs).
//
// This is the reference's brute-force intersector (tracer_base.py:168-278) kept in its own
// iteration order — objects in scene order, a per-object slab cull, then that object's
// primitives — but executed wave-wide:
// the object / primitive loop counters are wave-uniform, so primitive records come in through
// scalar loads (SGPR operands, no LDS, no per-lane stack) and there is no traversal divergence;
// an object is skipped for the whole wave only when no lane's ray needs it.  Because the order
// and every arithmetic operation are the reference's, ties resolve exactly as upstream.
// Scene tables are immutable while a renderer exists, so the sweep reads them through the constant
// address space: with a wave-uniform index the backend then selects s_load_dwordx4 (SGPR operands,
// scalar cache) instead of a vector global load per lane.
typedef const float __attribute__((address_space(4))) * cf_ptr;
typedef const int __attribute__((address_space(4))) * ci_ptr;
// Two triangles per lane per instruction: gfx950 issues v_pk_mul_f32 / v_pk_add_f32 at the rate of their
// scalar forms, and a packed multiply or add rounds each half exactly like the scalar instruction, so testing
// triangles k and k+1 in the two halves of a 64-bit register pair halves the VALU work without touching a bit
// of the result.  (No packed FMA is formed: -ffp-contract=off applies to vectors too.)
APT_D float4 ld4c(cf_ptr p) { return make_float4(p[0], p[1], p[2], p[3]); }
APT_D v2f ld2c(cf_ptr p) { v2f r; r.x = p[0]; r.y = p[1]; return r; }

// Sweep stream (built in apt_scene_create), one block per object in scene order, 8-float aligned:
//   [ (lo.x hi.x) (lo.y hi.y) (lo.z hi.z) 0 0 ]                        object slab bounds, interleaved
//   sphere   : [ centre.xyz r 0 0 0 0 ]
//   triangles: ceil(count/2) pair records of 24 floats, every quantity stored as (tri 2j, tri 2j+1):
//              p0.x p0.y p0.z  e1.x e1.y e1.z  e2.x e2.y e2.z  n.x n.y n.z      n = t-row cofactors of [e1 e2 .]
//              an odd tail is padded with an all-zero triangle (det = 0 -> NaN barycentrics -> never accepted)
struct SweepScene {
    const float* stream;
    const int* obj_tab;       // n_objects * 4: block offset (floats), primitive count, is_sphere, first primitive
    const int* prim_obj;      // primitive -> object
    int n_objects;
};

// per-lane state shared by the object loop: packed broadcasts of the ray
struct SweepRay {
    v2f ox, oy, oz, a02, a12, a22;
    APT_D void set(f3 o, f3 d) { ox = sp2(o.x); oy = sp2(o.y); oz = sp2(o.z); a02 = sp2(-d.x); a12 = sp2(-d.y); a22 = sp2(-d.z); }
};

// Per-object slab cull (TracerBase.aabb_test, tracer_base.py:159-166 and the `t_near > min_depth` skip at
// :184).  It is part of the result, not just a speed-up: a ray with a zero direction component whose origin
// sits on (or a rounding error outside) a slab boundary is culled upstream even though the triangle test
// would accept it.  The reference divides by the ray; dividing six times per object costs more than testing
// a triangle, so a reciprocal-multiply version with an error band decides the clear cases and only lanes
// inside the band (or with non-finite slabs) take the division path.
APT_D bool object_cull(v2f bound_x, v2f bound_y, v2f bound_z, const SweepRay& r, f3 d, f3 inv_d, float t_best, float& t_entry) {
    const v2f bx = bound_x - r.ox, by = bound_y - r.oy, bz = bound_z - r.oz;                     // (lo - o, hi - o) per axis
    v2f tx = bx * sp2(inv_d.x), ty = by * sp2(inv_d.y), tz = bz * sp2(inv_d.z);
    float tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fminf(tz.x, tz.y));
    float tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fmaxf(tz.x, tz.y));
    const float band = 2e-6f * (fabsf(tn) + fabsf(tf));
    const bool pass = (tn + band < tf) && (tf > 1e-30f) && (tn + band < t_best);
    const bool fail = (tn > tf + band) || (tf < 0.f) || (tn > t_best + band);
    bool ok = pass;
    if (!(pass || fail)) {
        tx.x = bx.x / d.x; tx.y = bx.y / d.x; ty.x = by.x / d.y; ty.y = by.y / d.y; tz.x = bz.x / d.z; tz.y = bz.y / d.z;
        tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fminf(tz.x, tz.y));
        tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fmaxf(tz.x, tz.y));
        ok = (tn < tf) && tf > 0.f && !(tn > t_best);
    }
    t_entry = tn;
    return ok;
}
APT_D bool object_cull(cf_ptr blk, const SweepRay& r, f3 d, f3 inv_d, float t_best) { float tn; return object_cull(ld2c(blk), ld2c(blk + 2), ld2c(blk + 4), r, d, inv_d, t_best, tn); }

// sphere record against the ray (tracer_base.py:184-199)
template <bool ANY>
APT_D void sphere_test(cf_ptr rec4, int first, f3 o, f3 d, bool need, bool& found, HitRec& rec) {
    float4 q0 = ld4c(rec4);
    f3 s2c = mk3(q0.x, q0.y, q0.z) - o;
    float r2 = q0.w * q0.w;
    float cn2 = norm2(s2c), proj = dot(d, s2c);
    float c2ray = cn2 - proj * proj;
    float cut = sqrtf(r2 - c2ray);
    float t = proj + ((cn2 > r2 + 1e-4f) ? -cut : cut);
    if (need && c2ray < r2 && t > 1e-4f && t < rec.t) {
        if (ANY) found = true;
        else { rec.t = t; rec.prim = first; rec.u = 0.f; rec.v = 0.f; }
    }
}

// one pair record in SGPRs
struct PairRec { v2f p0x, p0y, p0z, a00, a10, a20, a01, a11, a21, nx, ny, nz; };
APT_D PairRec ld_pair(cf_ptr r) {
    PairRec p;
    p.p0x = ld2c(r); p.p0y = ld2c(r + 2); p.p0z = ld2c(r + 4);
    p.a00 = ld2c(r + 6); p.a10 = ld2c(r + 8); p.a20 = ld2c(r + 10);
    p.a01 = ld2c(r + 12); p.a11 = ld2c(r + 14); p.a21 = ld2c(r + 16);
    p.nx = ld2c(r + 18); p.ny = ld2c(r + 20); p.nz = ld2c(r + 22);
    return p;
}
// all pair records of one mesh object against the ray, in primitive order
template <bool ANY>
APT_D void pair_tests(cf_ptr recs, int count, int first, const SweepRay& s, bool& need, bool& found, HitRec& rec) {
    const int n_pairs = (count + 1) >> 1;
    for (int j = 0; j < n_pairs; j++) {
        const PairRec p = ld_pair(recs + 24 * j);
        const v2f c00 = p.a11 * s.a22 - p.a21 * s.a12, c01 = p.a21 * s.a02 - p.a01 * s.a22, c02 = p.a01 * s.a12 - p.a11 * s.a02;
        const v2f det = (p.a00 * c00 + p.a10 * c01) + p.a20 * c02;
        v2f inv_det; inv_det.x = 1.0f / det.x; inv_det.y = 1.0f / det.y;
        const v2f sx = s.ox - p.p0x, sy = s.oy - p.p0y, sz = s.oz - p.p0z;
        const v2f c10 = s.a12 * p.a20 - s.a22 * p.a10, c11 = s.a22 * p.a00 - s.a02 * p.a20, c12 = s.a02 * p.a10 - s.a12 * p.a00;
        const v2f u = ((inv_det * c00) * sx + (inv_det * c01) * sy) + (inv_det * c02) * sz;
        const v2f v = ((inv_det * c10) * sx + (inv_det * c11) * sy) + (inv_det * c12) * sz;
        const v2f t = ((inv_det * p.nx) * sx + (inv_det * p.ny) * sy) + (inv_det * p.nz) * sz;
        const v2f uv = u + v;
        // triangle 2j, then 2j+1: the reference's sequential `t < min_depth` update order
        if (need && u.x >= 0.f && v.x >= 0.f && uv.x <= 1.0f && t.x > 1e-4f && t.x < rec.t) {
            if (ANY) { found = true; need = false; }
            else { rec.t = t.x; rec.prim = first + 2 * j; rec.u = u.x; rec.v = v.x; }
        }
        if (need && u.y >= 0.f && v.y >= 0.f && uv.y <= 1.0f && t.y > 1e-4f && t.y < rec.t) {
            if (ANY) { found = true; need = false; }
            else { rec.t = t.y; rec.prim = first + 2 * j + 1; rec.u = u.y; rec.v = v.y; }
        }
    }
}

// Wave-level sweep (unit-test entry kernels and anything launched without the workgroup scratch)
template <bool ANY>
APT_D bool sweep(const SweepScene& sc, f3 o, f3 d, HitRec& rec) {
    bool found = false;                                   // ANY: this lane already has its answer
    const cf_ptr stream = (cf_ptr)sc.stream;
    const ci_ptr tab = (ci_ptr)sc.obj_tab;
    const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    SweepRay sr; sr.set(o, d);
    for (int ob = 0; ob < sc.n_objects; ob++) {
        const cf_ptr blk = stream + tab[4 * ob];
        const int count = tab[4 * ob + 1], first = tab[4 * ob + 3];
        bool need = !found && object_cull(blk, sr, d, inv_d, rec.t);
        if (!__any(need)) continue;                       // wave-uniform skip
        if (tab[4 * ob + 2]) sphere_test<ANY>(blk + 8, first, o, d, need, found, rec);
        else pair_tests<ANY>(blk + 8, count, first, sr, need, found, rec);
        if (ANY && __all(found)) break;
    }
    return found;
}

// Workgroup-cooperative sweep.  A wave's 64 rays are incoherent after the first bounce, so the wave-uniform
// skip above almost never fires for an object like a Cornell block even though only ~1/4 of the lanes pass its
// slab cull.  For objects with APT_SWEEP_LIST_MIN or more primitives the block therefore gathers the lanes that
// DO need the object into an LDS work list and its waves run the pair records over dense 64-ray chunks of that
// list (ray and running-best state go through LDS).  Everything a ray sees is unchanged — objects in scene order,
// the cull against its own running t, primitives in order — so the hit is bit-identical to sweep().
// Must be called by every thread of the block (barriers inside); `lds` = APT_SWEEP_LDS_FLOATS floats.
#define APT_SWEEP_LIST_MIN 6
#define APT_SWEEP_MAX_LISTS 128
#define APT_SWEEP_LDS_FLOATS(block) (11 * (block) + APT_SWEEP_MAX_LISTS)
template <bool ANY, int NT>
APT_D bool sweep_wg(const SweepScene& sc, f3 o, f3 d, HitRec& rec, bool active, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* s_ray = lds;                                    // o.xyz d.xyz, SoA over the block
    float* s_t = lds + 6 * NT; int* s_prim = (int*)(lds + 7 * NT); float* s_u = lds + 8 * NT; float* s_v = lds + 9 * NT;
    int* s_list = (int*)(lds + 10 * NT); int* s_count = (int*)(lds + 11 * NT);
    bool found = false, staged = false;
    int n_lists = 0;
    const cf_ptr stream = (cf_ptr)sc.stream;
    const ci_ptr tab = (ci_ptr)sc.obj_tab;
    const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    SweepRay sr; sr.set(o, d);
    for (int ob = 0; ob < sc.n_objects; ob++) {
        const cf_ptr blk = stream + tab[4 * ob];
        const int count = tab[4 * ob + 1], first = tab[4 * ob + 3], is_sphere = tab[4 * ob + 2];
        bool need = active && !found && object_cull(blk, sr, d, inv_d, rec.t);
        if (is_sphere || count < APT_SWEEP_LIST_MIN || n_lists >= APT_SWEEP_MAX_LISTS) {
            if (!__any(need)) continue;
            if (is_sphere) sphere_test<ANY>(blk + 8, first, o, d, need, found, rec);
            else pair_tests<ANY>(blk + 8, count, first, sr, need, found, rec);
            continue;
        }
        if (!staged) {
            __syncthreads();                               // scratch may still be read by the previous tile
            s_ray[tid] = o.x; s_ray[NT + tid] = o.y; s_ray[2 * NT + tid] = o.z;
            s_ray[3 * NT + tid] = d.x; s_ray[4 * NT + tid] = d.y; s_ray[5 * NT + tid] = d.z;
            if (tid < APT_SWEEP_MAX_LISTS) s_count[tid] = 0;
            __syncthreads();
            staged = true;
        }
        {   // gather the lanes that need this object
            const unsigned long long m = __ballot(need);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(&s_count[n_lists], (int)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);      // v_readlane: lane 0's value as a scalar (a shuffle would go through the LDS crossbar)
            if (need) {
                s_list[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = tid;
                s_t[tid] = rec.t; s_prim[tid] = -1;
            }
        }
        __syncthreads();
        const int L = s_count[n_lists];
        for (int c = wave; c * 64 < L; c += NT / 64) {
            const int li = c * 64 + lane;
            bool has = li < L;
            const int ti = s_list[has ? li : L - 1];
            const f3 ro = mk3(s_ray[ti], s_ray[NT + ti], s_ray[2 * NT + ti]);
            const f3 rd = mk3(s_ray[3 * NT + ti], s_ray[4 * NT + ti], s_ray[5 * NT + ti]);
            SweepRay s2; s2.set(ro, rd);
            HitRec r2; r2.t = s_t[ti]; r2.prim = -1; r2.u = r2.v = 0.f;
            bool got = false;
            const bool mine = has;
            pair_tests<ANY>(blk + 8, count, first, s2, has, got, r2);
            if (ANY) { if (mine && got) s_prim[ti] = 1; }
            else if (mine && r2.prim >= 0) { s_t[ti] = r2.t; s_prim[ti] = r2.prim; s_u[ti] = r2.u; s_v[ti] = r2.v; }
        }
        __syncthreads();
        if (need && s_prim[tid] >= 0) {
            if (ANY) found = true;
            else { rec.t = s_t[tid]; rec.prim = s_prim[tid]; rec.u = s_u[tid]; rec.v = s_v[tid]; }
        }
        n_lists++;
    }
    return found;
}

// Tiled sweep: the block's NT rays against every object, with no wave ever testing a primitive for a lane
// that does not need it.
//
// The reference loop is sequential per ray (cull against the running t, then `t < min_depth` updates), but its
// RESULT is an order-independent reduction except on near-ties:  let t_i be object i's own first-minimal hit
// below the initial bound t0, tn_i its slab entry.  Upstream accepts i when !(tn_i > cur) and t_i < cur, cur being
// the minimum accepted so far.  Since a hit lies inside the box, tn_i <= t_i up to rounding, so `tn_i > cur`
// implies `t_i >= cur` unless cur falls in the rounding gap between t_i and tn_i — which needs a second surface
// within that gap.  Hence:  winner = min over (t_i, primitive index)  [strict `<` keeps the first of equal t],
// and any ray where (a) two candidates come within TIE_EPS of each other at the running minimum, or (b) a
// candidate's t_i undershoots its tn_i by more than TIE_EPS, is re-done by its owner with the sequential sweep.
// That makes the result exact in every case while the bulk runs as independent (ray, object) work items:
//   phase A  every thread culls its ray against every object (static bound t0) and appends itself to the LDS
//            list of each object it needs;
//   phase B  the block's waves walk all lists in dense 64-entry chunks (object wave-uniform -> scalar record
//            loads), reduce with a 64-bit LDS atomicMin on (t bits, primitive, list slot) and leave (u, v) in
//            the entry's slot;
//   phase C  the owner picks up its minimum (or its occlusion flag), or runs the fallback.
// LDS: NT * 40 B + n_objects * (NT * 8 B + 4 B); any-hit: NT * 32 B + n_objects * (NT * 2 B + 4 B).  Must be called by every thread of the block.
#define APT_TILE_MAX_OBJECTS 48
#define APT_TILE_LDS_BYTES(nt, n_obj) ((size_t)(nt) * 40 + (size_t)(n_obj) * ((size_t)(nt) * 8 + 4) + 16)
// any-hit tiles need neither the (t, primitive) reduction slots nor (tn | u, v) per entry: 2-byte entries
#define APT_TILE_LDS_BYTES_ANY(nt, n_obj) ((size_t)(nt) * 32 + (size_t)(n_obj) * ((size_t)(nt) * 2 + 4) + 16)
struct TileEntry { uint32_t a; float b; };                 // (ray, tn) going in, (u, v) coming out
template <bool ANY, int NT>
APT_D bool sweep_tile(const SweepScene& sc, f3 o, f3 d, HitRec& rec, bool active, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_obj = sc.n_objects;
    float* s_ray = lds;                                                     // o.xyz d.xyz t0, SoA
    int* s_flag = reinterpret_cast<int*>(lds + 7 * NT);
    unsigned long long* s_best = reinterpret_cast<unsigned long long*>(lds + 8 * NT);    // 8-byte aligned
    TileEntry* s_ent = reinterpret_cast<TileEntry*>(lds + 10 * NT);          // closest hit: [object][slot] (ray, tn) -> (u, v)
    uint16_t* s_idx = reinterpret_cast<uint16_t*>(lds + 8 * NT);            // any hit: [object][slot] ray only (overlays s_best, unused there)
    int* s_cnt = ANY ? reinterpret_cast<int*>(s_idx + (size_t)n_obj * NT) : reinterpret_cast<int*>(s_ent + (size_t)n_obj * NT);
    const float t0 = rec.t;
    const cf_ptr stream = (cf_ptr)sc.stream;
    const ci_ptr tab = (ci_ptr)sc.obj_tab;
    s_ray[tid] = o.x; s_ray[NT + tid] = o.y; s_ray[2 * NT + tid] = o.z;
    s_ray[3 * NT + tid] = d.x; s_ray[4 * NT + tid] = d.y; s_ray[5 * NT + tid] = d.z; s_ray[6 * NT + tid] = t0;
    s_flag[tid] = 0;
    if (!ANY) s_best[tid] = ~0ull;
    if (tid < n_obj) s_cnt[tid] = 0;
    __syncthreads();
    {   // phase A
        const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        SweepRay sr; sr.set(o, d);
        for (int ob = 0; ob < n_obj; ob++) {
            const cf_ptr blk = stream + tab[4 * ob];
            float tn;
            const bool need = object_cull(ld2c(blk), ld2c(blk + 2), ld2c(blk + 4), sr, d, inv_d, t0, tn) && active;
            const unsigned long long m = __ballot(need);
            if (!m) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt[ob], (int)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (need) {
                const int at = ob * NT + base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (ANY) s_idx[at] = (uint16_t)tid;
                else { TileEntry e; e.a = (uint32_t)tid; e.b = tn; s_ent[at] = e; }
            }
        }
    }
    __syncthreads();
    {   // phase B: global chunk g of the concatenated lists -> (object, chunk)
        int g = wave;
        for (int ob = 0; ob < n_obj; ob++) {
            const int L = s_cnt[ob], n_chunks = (L + 63) >> 6;
            if (g >= n_chunks) { g -= n_chunks; continue; }
            const cf_ptr blk = stream + tab[4 * ob];
            const int count = tab[4 * ob + 1], first = tab[4 * ob + 3], is_sphere = tab[4 * ob + 2];
            for (; g < n_chunks; g += NT / 64) {
                const int li = g * 64 + lane;
                const bool has = li < L;
                const int at = ob * NT + (has ? li : L - 1);
                TileEntry* slot = s_ent + at;
                TileEntry e; e.a = 0u; e.b = 0.f;
                if (ANY) e.a = s_idx[at]; else e = *slot;
                const int ti = (int)e.a;
                const f3 ro = mk3(s_ray[ti], s_ray[NT + ti], s_ray[2 * NT + ti]);
                const f3 rd = mk3(s_ray[3 * NT + ti], s_ray[4 * NT + ti], s_ray[5 * NT + ti]);
                HitRec r2; r2.t = s_ray[6 * NT + ti]; r2.prim = -1; r2.u = r2.v = 0.f;
                bool nd = has, got = false;
                if (is_sphere) sphere_test<ANY>(blk + 8, first, ro, rd, nd, got, r2);
                else { SweepRay s2; s2.set(ro, rd); pair_tests<ANY>(blk + 8, count, first, s2, nd, got, r2); }
                if (ANY) { if (has && got) s_flag[ti] = 1; }
                else if (has && r2.prim >= 0) {
                    const float eps = 1e-5f * r2.t + 1e-6f;
                    const unsigned long long key = ((unsigned long long)__float_as_uint(r2.t) << 32) | ((unsigned long long)(uint32_t)r2.prim << 16) | (unsigned long long)(uint32_t)li;
                    const unsigned long long old = atomicMin(&s_best[ti], key);
                    bool redo = e.b - r2.t > eps;
                    if (old != ~0ull) redo = redo || fabsf(__uint_as_float((uint32_t)(old >> 32)) - r2.t) <= eps;
                    if (redo) s_flag[ti] = 1;
                    TileEntry w; w.a = __float_as_uint(r2.u); w.b = r2.v; *slot = w;
                }
            }
            g -= n_chunks;
        }
    }
    __syncthreads();
    if (ANY) return s_flag[tid] != 0;
    if (s_flag[tid]) { rec.t = t0; rec.prim = -1; rec.u = rec.v = 0.f; sweep<false>(sc, o, d, rec); return false; }
    const unsigned long long key = s_best[tid];
    if (key != ~0ull) {
        rec.t = __uint_as_float((uint32_t)(key >> 32)); rec.prim = (int)((key >> 16) & 0xffffu);
        const TileEntry w = s_ent[sc.prim_obj[rec.prim] * NT + (int)(key & 0xffffu)];
        rec.u = __uint_as_float(w.a); rec.v = w.b;
    }
    return false;
}
APT_D bool sweep_any(const SweepScene& sc, f3 o, f3 d, HitRec& rec) { return sweep<true>(sc, o, d, rec); }

// ---------------------------------------------------------------------------------------------
// Flat sweep (fast build, scenes of up to APT_FLAT_MAX_PRIMS primitives: the Cornell configs).
//
// The exact build's small-scene modes above execute the reference's loop (tracer_base.py:168-278) operation for operation: slab cull
// per object, then the adjugate solve of [e1 e2 -d] per triangle - ~35 VALU instructions per triangle plus the culls, list building
// and barriers that keep lanes from testing objects they do not need.  Inside SURVEY 8(d)'s tolerance (t within 1e-5 relative, same
// object unless tied) the same hit costs a third of that:
//   * every planar primitive carries a precomputed affine map world -> (u, v, h) (rows U, V, T of [e1 e2 n]^-1 with the translation
//     folded in; Baldwin & Weber, JCGT 2016): t = -T(o) / T(d), u = U(o) + t U(d), v = V(o) + t V(d) - nine FMAs for the origin,
//     nine for the direction, ONE reciprocal;
//   * two triangles that form a parallelogram (every wall and box face of the Cornell scenes) are ONE record: inside <=> u, v in [0, 1];
//     which of the two triangles was hit, and its own barycentrics, are decided once per ray from the winner's (u, v);
//   * TWO RAYS PER LANE: a lane owns queue entries 2k and 2k + 1, which arrive as one 8-byte load per component straight into the
//     register pair a packed instruction wants (v_pk_fma_f32: ray A in the low half, ray B in the high half), while the record is a
//     wave-uniform scalar-load operand broadcast to both halves.  A wave therefore tests 128 rays per record fetch, has twice the
//     loads in flight per round trip (the stage is latency-bound: a few hundred instructions between dependent memory accesses), and
//     pays its loop control, address arithmetic and scalar traffic once per 128 rays;
//   * no culls, no lists, no LDS, no barriers: a ray's work is the scene's record count, the same for every lane.
// The one place where the reference's per-object slab cull decides a result - 0 / 0 = NaN slabs of rays with a zero direction
// component whose origin lies on a box plane (tracer_base.py:159-166; DESIGN.md "the per-object slab cull is part of the result") - is
// kept: such rays re-run the reference-order sweep() above after the flat loop.
#ifndef APT_FLAT_MAX_PRIMS
#define APT_FLAT_MAX_PRIMS 96
#endif
struct FlatScene {
    const float* stream;      // [parallelograms][same, in a coplanar group] x 12 floats (corner p0, rows U, V, T) | [convex quads][same, coplanar group] x 18 (+ the two far edges' functions of (u, v)) | [triangles][same, coplanar group] x 12 | [spheres] x 4 (centre, r^2)
    const float4* tab;        // per record, 7 float4: (U, p0.x) (V, p0.y) | prim_a prim_b class_a class_b | map_a (u0 uu uv v0 vu vv) map_b (same) | (p0.z, -, -, -)
    const float* precom;      // n_prims * 9: (e1, e2, p0) per triangle - the reference's own test decides between near-tied coplanar candidates
    const float* pairs;       // the same records two by two, every float of records 2j and 2j + 1 next to each other (an odd tail repeats its last record): the any-hit sweep of ONE ray per lane tests two records per packed instruction (flat_any1: shade kernels that trace their own light samples)
    int n_quads, n_quads_tie, n_gquads, n_gquads_tie, n_tris, n_tris_tie, n_spheres;
    int defer_all;            // test switch (APT_FLAT_DEFER_ALL=1 at scene creation): every ray takes the reference-order path - in the stage kernels, through the fix-up lists
};
#if APT_FAST
struct FlatRays { v2f ox, oy, oz, dx, dy, dz; };           // two rays: .x = entry 2k, .y = entry 2k + 1
// (t, u, v) of both rays against one planar record: 12 wave-uniform floats at r = corner p0, rows U, V, T of [e1 e2 n]^-1:
// t = -T.s / T.d, then u = U.P, v = V.P at P = s + t d.
// The origin enters as s = o - p0, exactly the reference's first operation (tracer_base.py:206): for a ray that STARTS on the primitive
// (every continuation and shadow ray does, on one) the height T . s is then a sum of small products, and its rounding noise - which decides
// whether a grazing ray re-hits its own surface beyond the 1e-4 threshold - stays at the reference's level.  (With the translation folded
// into a fourth column the terms were of the size of the coordinates: twice the noise, +0.3 % path vertices on scenes/test/textured.xml.)
APT_D void planar_solve(cf_ptr r, const FlatRays& q, v2f& t, v2f& u, v2f& v) {
    const v2f sx = q.ox - sp2(r[0]), sy = q.oy - sp2(r[1]), sz = q.oz - sp2(r[2]);
    const v2f ux = sp2(r[3]), uy = sp2(r[4]), uz = sp2(r[5]);
    const v2f vx = sp2(r[6]), vy = sp2(r[7]), vz = sp2(r[8]);
    const v2f tx = sp2(r[9]), ty = sp2(r[10]), tz = sp2(r[11]);
    const v2f t_o = fma2(tx, sx, fma2(ty, sy, tz * sz));
    const v2f t_d = fma2(tx, q.dx, fma2(ty, q.dy, tz * q.dz));
    v2f inv; inv.x = __builtin_amdgcn_rcpf(t_d.x); inv.y = __builtin_amdgcn_rcpf(t_d.y);
    t = -t_o * inv;
    // (u, v) from the hit point relative to the corner, P = s + t d - what flat_resolve() does for the winner, so the inside test and the
    // reported barycentrics are one formula - 9 packed operations instead of the 14 of U.s + t (U.d), V.s + t (V.d)
    const v2f px = fma2(t, q.dx, sx), py = fma2(t, q.dy, sy), pz = fma2(t, q.dz, sz);
    u = fma2(ux, px, fma2(uy, py, uz * pz));
    v = fma2(vx, px, fma2(vy, py, vz * pz));
}
struct FlatHit2 { v2f t; int idx0, idx1, run0, run1; };   // closest record per ray (-1: none), its distance (or the search limit), and a near-tied runner-up (-1: none)
// Coplanar primitives (a glass box resting on the floor): which of two faces at the SAME distance a ray "hits" is decided upstream by
// the last bit of two separately rounded distances and the strict `t < min_depth` of its loop (tracer_base.py:208) - the earlier primitive
// wins unless the later one's t rounds lower.  The records of such groups (found at scene creation: identical plane, overlapping extent)
// sit in sections of their own whose loop also remembers a
// candidate within FLAT_TIE_EPS of the running minimum; flat_tie_break() then lets the reference's own arithmetic decide between the two.
#define FLAT_TIE_REL 1e-5f
#define FLAT_TIE_ABS 1e-6f
struct FlatBest { float t; int idx, runner; };               // running closest hit of one ray: distance (or the search limit), record (-1: none), near-tied runner-up (-1: none)
// (passed and returned BY VALUE: with reference parameters the backend kept the four indices of a lane in scratch memory)
template <bool TIE>
APT_D FlatBest flat_candidate(FlatBest b, bool inside, float t, int idx) {
    const bool valid = inside && t > 1e-4f;
    if (TIE) {
        const float gap = FLAT_TIE_REL * b.t + FLAT_TIE_ABS;
        const bool better = valid && t < b.t, near_behind = valid && !better && (t - b.t <= gap);
        const int run_new = (b.t - t <= gap) ? b.idx : -1;
        b.runner = better ? run_new : (near_behind ? idx : b.runner);
        b.idx = better ? idx : b.idx;
        b.t = better ? t : b.t;
    } else {
        const bool better = valid && t < b.t;
        b.idx = better ? idx : b.idx;
        b.t = better ? t : b.t;
    }
    return b;
}
APT_D bool flat_blocks(bool inside, float t, float lim) { return inside && t > 1e-4f && t < lim; }
// Both rays against every record.  ANY = false: closest hit below `lim` per ray.  ANY = true: occ0 / occ1 = something lies in (1e-4, lim).
template <bool ANY>
APT_D void flat_loop(const FlatScene& fl, const FlatRays& q, v2f lim, FlatHit2& h, bool& occ0, bool& occ1) {
    cf_ptr at = (cf_ptr)fl.stream;
    FlatBest x, y; x.t = lim.x; y.t = lim.y; x.idx = y.idx = -1; x.runner = y.runner = -1;
    bool o0 = false, o1 = false;
    int idx = 0;                                           // wave-uniform record index
    // parallelograms: inside <=> |u - 1/2| <= 1/2 and |v - 1/2| <= 1/2
    const int nq_plain = ANY ? fl.n_quads + fl.n_quads_tie : fl.n_quads;
    for (int j = 0; j < nq_plain; j++, idx++, at += 12) {
        v2f t, u, v; planar_solve(at, q, t, u, v);
        const v2f a = u - sp2(0.5f), b = v - sp2(0.5f);
        const bool i0 = fmaxf(fabsf(a.x), fabsf(b.x)) <= 0.5f, i1 = fmaxf(fabsf(a.y), fabsf(b.y)) <= 0.5f;
        if (ANY) { o0 = o0 || flat_blocks(i0, t.x, lim.x); o1 = o1 || flat_blocks(i1, t.y, lim.y); }
        else { x = flat_candidate<false>(x, i0, t.x, idx); y = flat_candidate<false>(y, i1, t.y, idx); }
    }
    if (!ANY) for (int j = 0; j < fl.n_quads_tie; j++, idx++, at += 12) {
        v2f t, u, v; planar_solve(at, q, t, u, v);
        const v2f a = u - sp2(0.5f), b = v - sp2(0.5f);
        x = flat_candidate<true>(x, fmaxf(fabsf(a.x), fabsf(b.x)) <= 0.5f, t.x, idx);
        y = flat_candidate<true>(y, fmaxf(fabsf(a.y), fabsf(b.y)) <= 0.5f, t.y, idx);
    }
    // convex quadrilaterals (two coplanar triangles sharing an edge that do NOT close a parallelogram: the slanted faces of a sheared box):
    // in the frame of the first triangle (corner opposite the shared edge) the outline is u >= 0, v >= 0 and the two far edges, each an
    // affine function of (u, v) stored behind the rows: inside <=> min(u, v, e1, e2) >= 0.  One plane solve instead of two.
    const int ng_plain = ANY ? fl.n_gquads + fl.n_gquads_tie : fl.n_gquads;
    for (int j = 0; j < ng_plain; j++, idx++, at += 18) {
        v2f t, u, v; planar_solve(at, q, t, u, v);
        const v2f e1 = fma2(sp2(at[12]), u, fma2(sp2(at[13]), v, sp2(at[14]))), e2 = fma2(sp2(at[15]), u, fma2(sp2(at[16]), v, sp2(at[17])));
        const bool i0 = fminf(fminf(u.x, v.x), fminf(e1.x, e2.x)) >= 0.f, i1 = fminf(fminf(u.y, v.y), fminf(e1.y, e2.y)) >= 0.f;
        if (ANY) { o0 = o0 || flat_blocks(i0, t.x, lim.x); o1 = o1 || flat_blocks(i1, t.y, lim.y); }
        else { x = flat_candidate<false>(x, i0, t.x, idx); y = flat_candidate<false>(y, i1, t.y, idx); }
    }
    if (!ANY) for (int j = 0; j < fl.n_gquads_tie; j++, idx++, at += 18) {
        v2f t, u, v; planar_solve(at, q, t, u, v);
        const v2f e1 = fma2(sp2(at[12]), u, fma2(sp2(at[13]), v, sp2(at[14]))), e2 = fma2(sp2(at[15]), u, fma2(sp2(at[16]), v, sp2(at[17])));
        x = flat_candidate<true>(x, fminf(fminf(u.x, v.x), fminf(e1.x, e2.x)) >= 0.f, t.x, idx);
        y = flat_candidate<true>(y, fminf(fminf(u.y, v.y), fminf(e1.y, e2.y)) >= 0.f, t.y, idx);
    }
    // triangles: inside <=> min(u, v, 1 - u - v) >= 0
    const int nt_plain = ANY ? fl.n_tris + fl.n_tris_tie : fl.n_tris;
    for (int j = 0; j < nt_plain; j++, idx++, at += 12) {
        v2f t, u, v; planar_solve(at, q, t, u, v);
        const v2f w = (sp2(1.0f) - u) - v;
        const bool i0 = fminf(fminf(u.x, v.x), w.x) >= 0.f, i1 = fminf(fminf(u.y, v.y), w.y) >= 0.f;
        if (ANY) { o0 = o0 || flat_blocks(i0, t.x, lim.x); o1 = o1 || flat_blocks(i1, t.y, lim.y); }
        else { x = flat_candidate<false>(x, i0, t.x, idx); y = flat_candidate<false>(y, i1, t.y, idx); }
    }
    if (!ANY) for (int j = 0; j < fl.n_tris_tie; j++, idx++, at += 12) {
        v2f t, u, v; planar_solve(at, q, t, u, v);
        const v2f w = (sp2(1.0f) - u) - v;
        x = flat_candidate<true>(x, fminf(fminf(u.x, v.x), w.x) >= 0.f, t.x, idx);
        y = flat_candidate<true>(y, fminf(fminf(u.y, v.y), w.y) >= 0.f, t.y, idx);
    }
    // spheres: the reference's test operation for operation (tracer_base.py:184-199; un-fused products, IEEE square root) - a grazing hit is a
    // difference of two nearly equal squares, and on the mirror and glass balls of the Cornell scenes every digit of it is amplified by
    // the bounces that follow; with the same ray the distance is the exact build's, bit for bit.  Record = centre.xyz, r^2.
    for (int j = 0; j < fl.n_spheres; j++, idx++, at += 4) {
        const v2f r2 = sp2(at[3]);
        const v2f sx = sp2(at[0]) - q.ox, sy = sp2(at[1]) - q.oy, sz = sp2(at[2]) - q.oz;
        const v2f cn2 = (sx * sx + sy * sy) + sz * sz;
        const v2f proj = (q.dx * sx + q.dy * sy) + q.dz * sz;
        const v2f c2ray = cn2 - proj * proj;
        const v2f disc = r2 - c2ray;
        const float cut0 = sqrtf(disc.x), cut1 = sqrtf(disc.y);
        const float ta = proj.x + ((cn2.x > r2.x + 1e-4f) ? -cut0 : cut0), tb = proj.y + ((cn2.y > r2.y + 1e-4f) ? -cut1 : cut1);
        const bool i0 = c2ray.x < r2.x, i1 = c2ray.y < r2.y;
        if (ANY) { o0 = o0 || flat_blocks(i0, ta, lim.x); o1 = o1 || flat_blocks(i1, tb, lim.y); }
        else { x = flat_candidate<false>(x, i0, ta, idx); y = flat_candidate<false>(y, i1, tb, idx); }
    }
    // A runner-up only means something next to a winner from a coplanar-group section: the plain sections that run after a tie section
    // (convex quads, triangles, spheres) replace the winner without looking at the runner, so a sphere or a plain face in front of a coplanar
    // pair would otherwise end with the pair's stale runner and be "tie-broken" against a face far behind it.  Checked once per ray here
    // instead of once per record in flat_candidate<false> (the hot loop of scenes without any coplanar group stays as it is).
    if (!ANY && (fl.n_quads_tie | fl.n_gquads_tie | fl.n_tris_tie) != 0) {
        const int a0 = fl.n_quads, a1 = a0 + fl.n_quads_tie, b0 = a1 + fl.n_gquads, b1 = b0 + fl.n_gquads_tie, c0 = b1 + fl.n_tris, c1 = c0 + fl.n_tris_tie;
        const bool tx = (x.idx >= a0 && x.idx < a1) || (x.idx >= b0 && x.idx < b1) || (x.idx >= c0 && x.idx < c1);
        const bool ty = (y.idx >= a0 && y.idx < a1) || (y.idx >= b0 && y.idx < b1) || (y.idx >= c0 && y.idx < c1);
        x.runner = tx ? x.runner : -1; y.runner = ty ? y.runner : -1;
    }
    h.t = mk2(x.t, y.t); h.idx0 = x.idx; h.idx1 = y.idx; h.run0 = x.runner; h.run1 = y.runner;
    occ0 = o0; occ1 = o1;
}
// the winning record's own triangle, barycentrics and material class: hit point -> (u, v) of the record, then the triangle's affine map
APT_D void flat_resolve(const FlatScene& fl, int idx, float t, f3 o, f3 d, HitRec& rec, int& cls) {
    const float4* e = fl.tab + 7 * idx;
    const float4 U = e[0], V = e[1], ids = e[2], ma0 = e[3], mab = e[4], mb1 = e[5], pz = e[6];      // U.w, V.w, pz.x = the record's corner p0
    const f3 P = mk3(__builtin_fmaf(t, d.x, o.x - U.w), __builtin_fmaf(t, d.y, o.y - V.w), __builtin_fmaf(t, d.z, o.z - pz.x));      // s + t d, as planar_solve()
    const float u = __builtin_fmaf(U.x, P.x, __builtin_fmaf(U.y, P.y, U.z * P.z));
    const float v = __builtin_fmaf(V.x, P.x, __builtin_fmaf(V.y, P.y, V.z * P.z));
    const int prim_b = __float_as_int(ids.y);
    const bool second = prim_b >= 0 && u + v > 1.0f;
    const float m0 = second ? mab.z : ma0.x, m1 = second ? mab.w : ma0.y, m2 = second ? mb1.x : ma0.z;
    const float m3 = second ? mb1.y : ma0.w, m4 = second ? mb1.z : mab.x, m5 = second ? mb1.w : mab.y;
    rec.t = t; rec.prim = second ? prim_b : __float_as_int(ids.x);
    cls = second ? __float_as_int(ids.w) : __float_as_int(ids.z);
    rec.u = __builtin_fmaf(m1, u, __builtin_fmaf(m2, v, m0));
    rec.v = __builtin_fmaf(m4, u, __builtin_fmaf(m5, v, m3));
}
// Near-tied coplanar candidates: the reference's own triangle test (prim_test: the adjugate solve, un-fused) on the two triangles, taken in
// the reference's order (primitive index) with its strict `t < min_depth` - i.e. what upstream's loop would have kept.
APT_D void flat_tie_break(const FlatScene& fl, int idx_win, int idx_run, float t_flat, float lim, f3 o, f3 d, HitRec& rec, int& cls) {
    HitRec ra, rb; int ca, cb;
    const int n_planar = fl.n_quads + fl.n_quads_tie + fl.n_gquads + fl.n_gquads_tie + fl.n_tris + fl.n_tris_tie;
    if (idx_win >= n_planar || idx_run >= n_planar) { flat_resolve(fl, idx_win, t_flat, o, d, rec, cls); return; }     // (a sphere is never part of a coplanar group: flat_loop clears such runners; kept as a guard - prim_test below reads triangle records)
    flat_resolve(fl, idx_win, t_flat, o, d, ra, ca);
    flat_resolve(fl, idx_run, t_flat, o, d, rb, cb);          // (the runner-up's plane is the winner's: the same hit point picks its triangle)
    if (rb.prim < ra.prim) { const HitRec tr = ra; ra = rb; rb = tr; const int tc = ca; ca = cb; cb = tc; }     // ra: the earlier primitive
    float cur = lim; bool got = false;
    {
        const float* pc = fl.precom + 9 * ra.prim; float u, v;
        const float t = prim_test(make_float4(pc[6], pc[7], pc[8], pc[0]), make_float4(pc[1], pc[2], pc[3], pc[4]), make_float4(pc[5], 0.f, 0.f, 0.f), o, d, u, v);
        if (t > 1e-4f && t < cur) { cur = t; rec.t = t; rec.prim = ra.prim; rec.u = u; rec.v = v; cls = ca; got = true; }
    }
    {
        const float* pc = fl.precom + 9 * rb.prim; float u, v;
        const float t = prim_test(make_float4(pc[6], pc[7], pc[8], pc[0]), make_float4(pc[1], pc[2], pc[3], pc[4]), make_float4(pc[5], 0.f, 0.f, 0.f), o, d, u, v);
        if (t > 1e-4f && t < cur) { cur = t; rec.t = t; rec.prim = rb.prim; rec.u = u; rec.v = v; cls = cb; got = true; }
    }
    if (!got) flat_resolve(fl, idx_win, t_flat, o, d, rec, cls);     // both rejected by the exact test (an edge in the last bit): keep the flat answer
}
// Rays for which upstream's per-object slab cull (tracer_base.py:159-166,178-180) is part of the RESULT, not just a speed-up:
//  * a zero direction component: the cull may see 0 / 0 = NaN and skip an object the primitive test would accept;
//  * a direction that is not of unit length, in a scene with spheres: upstream's sphere test (tracer_base.py:184-199) assumes |d| = 1 and
//    reports hits a longer direction does not have - except where the sphere's box has already rejected the ray.  Such directions exist:
//    a normal map makes the shading normal, and with it the sampled direction, any length (scenes/test/textured.xml: |d| up to 1.2;
//    18 % more energy in that picture when the cull was left out).
// These rays re-run the reference-order sweep() after the flat loop.
APT_D bool flat_needs_cull(const FlatScene& fl, f3 d) {
    return d.x == 0.f || d.y == 0.f || d.z == 0.f || (fl.n_spheres > 0 && fabsf(((d.x * d.x + d.y * d.y) + d.z * d.z) - 1.0f) > 1e-4f) || fl.defer_all != 0;
}

// Closest hits of the lane's two rays (search limits in rec0.t / rec1.t); cls0 / cls1 = material class of the hit primitive (-1: miss).
// DEFER = true (the stage kernels' hot variant): the two rare cases whose answer needs the reference's own arithmetic - a near-tied
// coplanar runner-up, a ray for which upstream's slab cull is part of the result - are NOT settled here: sp0 / sp1 report them and the
// caller hands those entries to a follow-up launch (stages.hpp "fix-up lists").  Inlined, prim_test() and the reference-order sweep()
// set the register allocation of the whole kernel: k_extend_flat 79 -> 52 VGPRs, k_shadow_flat 85 -> 44 without them - and the register
// file, not issue slots or bandwidth, is what the three-lane pipeline runs out of.
template <bool DEFER>
APT_D void flat_closest2(const FlatScene& fl, const SweepScene& sw, const int* prim_class, f3 o0, f3 d0, f3 o1, f3 d1, HitRec& rec0, HitRec& rec1, int& cls0, int& cls1, bool& sp0, bool& sp1) {
    FlatRays q; q.ox = mk2(o0.x, o1.x); q.oy = mk2(o0.y, o1.y); q.oz = mk2(o0.z, o1.z); q.dx = mk2(d0.x, d1.x); q.dy = mk2(d0.y, d1.y); q.dz = mk2(d0.z, d1.z);
    FlatHit2 h; bool x0, x1;
    const float lim0 = rec0.t, lim1 = rec1.t;
    flat_loop<false>(fl, q, mk2(lim0, lim1), h, x0, x1);
    cls0 = -1; cls1 = -1;
    if (h.idx0 >= 0) flat_resolve(fl, h.idx0, h.t.x, o0, d0, rec0, cls0);
    if (h.idx1 >= 0) flat_resolve(fl, h.idx1, h.t.y, o1, d1, rec1, cls1);
    sp0 = sp1 = false;
    if (DEFER) { sp0 = h.run0 >= 0 || flat_needs_cull(fl, d0); sp1 = h.run1 >= 0 || flat_needs_cull(fl, d1); return; }
    if (__any(h.run0 >= 0 || h.run1 >= 0)) {                 // only scenes with reachable coplanar faces ever get here
        if (h.run0 >= 0) flat_tie_break(fl, h.idx0, h.run0, h.t.x, lim0, o0, d0, rec0, cls0);
        if (h.run1 >= 0) flat_tie_break(fl, h.idx1, h.run1, h.t.y, lim1, o1, d1, rec1, cls1);
    }
    const bool z0 = flat_needs_cull(fl, d0), z1 = flat_needs_cull(fl, d1);
    if (__any(z0 || z1)) {                                   // rare: directions like (0, 1, 0) come from degenerate samples only
        if (z0) { rec0.t = lim0; rec0.prim = -1; rec0.u = rec0.v = 0.f; sweep<false>(sw, o0, d0, rec0); cls0 = rec0.prim >= 0 ? prim_class[rec0.prim] : -1; }
        if (z1) { rec1.t = lim1; rec1.prim = -1; rec1.u = rec1.v = 0.f; sweep<false>(sw, o1, d1, rec1); cls1 = rec1.prim >= 0 ? prim_class[rec1.prim] : -1; }
    }
}
APT_D void flat_closest2(const FlatScene& fl, const SweepScene& sw, const int* prim_class, f3 o0, f3 d0, f3 o1, f3 d1, HitRec& rec0, HitRec& rec1, int& cls0, int& cls1) {
    bool a, b; flat_closest2<false>(fl, sw, prim_class, o0, d0, o1, d1, rec0, rec1, cls0, cls1, a, b);
}
// Occlusion of the lane's two rays below lim0 / lim1 (DEFER: see flat_closest2)
template <bool DEFER>
APT_D void flat_any2(const FlatScene& fl, const SweepScene& sw, f3 o0, f3 d0, f3 o1, f3 d1, float lim0, float lim1, bool& occ0, bool& occ1, bool& sp0, bool& sp1) {
    FlatRays q; q.ox = mk2(o0.x, o1.x); q.oy = mk2(o0.y, o1.y); q.oz = mk2(o0.z, o1.z); q.dx = mk2(d0.x, d1.x); q.dy = mk2(d0.y, d1.y); q.dz = mk2(d0.z, d1.z);
    FlatHit2 h;
    flat_loop<true>(fl, q, mk2(lim0, lim1), h, occ0, occ1);
    const bool z0 = flat_needs_cull(fl, d0), z1 = flat_needs_cull(fl, d1);
    sp0 = sp1 = false;
    if (DEFER) { sp0 = z0; sp1 = z1; return; }
    if (__any(z0 || z1)) {
        HitRec r; r.prim = -1; r.u = r.v = 0.f;
        if (z0) { r.t = lim0; occ0 = sweep<true>(sw, o0, d0, r); }
        if (z1) { r.t = lim1; occ1 = sweep<true>(sw, o1, d1, r); }
    }
}
APT_D void flat_any2(const FlatScene& fl, const SweepScene& sw, f3 o0, f3 d0, f3 o1, f3 d1, float lim0, float lim1, bool& occ0, bool& occ1) {
    bool a, b; flat_any2<false>(fl, sw, o0, d0, o1, d1, lim0, lim1, occ0, occ1, a, b);
}
// Occlusion of ONE ray per lane below `lim`: the lane's ray against two records per packed instruction (FlatScene::pairs) - the mirror
// image of flat_loop<true>, for kernels that hold one path per lane (the shade kernel that traces its own rays: shade_stage.hpp
// "rays traced in place").  Same arithmetic per (ray, record) as flat_loop: the two answers are the same bit for bit.
APT_D bool flat_any1(const FlatScene& fl, f3 o, f3 d, float lim) {
    cf_ptr at = (cf_ptr)fl.pairs;
    const v2f ox = sp2(o.x), oy = sp2(o.y), oz = sp2(o.z), dx = sp2(d.x), dy = sp2(d.y), dz = sp2(d.z);
    bool occ = false;
    auto solve = [&](cf_ptr r, v2f& t, v2f& u, v2f& v) {      // planar_solve() with the record pair in the halves and the ray broadcast
        const v2f sx = ox - ld2c(r), sy = oy - ld2c(r + 2), sz = oz - ld2c(r + 4);
        const v2f ux = ld2c(r + 6), uy = ld2c(r + 8), uz = ld2c(r + 10);
        const v2f vx = ld2c(r + 12), vy = ld2c(r + 14), vz = ld2c(r + 16);
        const v2f tx = ld2c(r + 18), ty = ld2c(r + 20), tz = ld2c(r + 22);
        const v2f t_o = fma2(tx, sx, fma2(ty, sy, tz * sz));
        const v2f t_d = fma2(tx, dx, fma2(ty, dy, tz * dz));
        v2f inv; inv.x = __builtin_amdgcn_rcpf(t_d.x); inv.y = __builtin_amdgcn_rcpf(t_d.y);
        t = -t_o * inv;
        const v2f px = fma2(t, dx, sx), py = fma2(t, dy, sy), pz = fma2(t, dz, sz);
        u = fma2(ux, px, fma2(uy, py, uz * pz));
        v = fma2(vx, px, fma2(vy, py, vz * pz));
    };
    const int nq = (fl.n_quads + fl.n_quads_tie + 1) >> 1, ng = (fl.n_gquads + fl.n_gquads_tie + 1) >> 1, nt = (fl.n_tris + fl.n_tris_tie + 1) >> 1, ns = (fl.n_spheres + 1) >> 1;
    for (int j = 0; j < nq; j++, at += 24) {
        v2f t, u, v; solve(at, t, u, v);
        const v2f a = u - sp2(0.5f), b = v - sp2(0.5f);
        occ = occ || flat_blocks(fmaxf(fabsf(a.x), fabsf(b.x)) <= 0.5f, t.x, lim) || flat_blocks(fmaxf(fabsf(a.y), fabsf(b.y)) <= 0.5f, t.y, lim);
    }
    for (int j = 0; j < ng; j++, at += 36) {
        v2f t, u, v; solve(at, t, u, v);
        const v2f e1 = fma2(ld2c(at + 24), u, fma2(ld2c(at + 26), v, ld2c(at + 28))), e2 = fma2(ld2c(at + 30), u, fma2(ld2c(at + 32), v, ld2c(at + 34)));
        occ = occ || flat_blocks(fminf(fminf(u.x, v.x), fminf(e1.x, e2.x)) >= 0.f, t.x, lim) || flat_blocks(fminf(fminf(u.y, v.y), fminf(e1.y, e2.y)) >= 0.f, t.y, lim);
    }
    for (int j = 0; j < nt; j++, at += 24) {
        v2f t, u, v; solve(at, t, u, v);
        const v2f w = (sp2(1.0f) - u) - v;
        occ = occ || flat_blocks(fminf(fminf(u.x, v.x), w.x) >= 0.f, t.x, lim) || flat_blocks(fminf(fminf(u.y, v.y), w.y) >= 0.f, t.y, lim);
    }
    for (int j = 0; j < ns; j++, at += 8) {                   // spheres: flat_loop's test (the reference's, tracer_base.py:184-199) on two spheres
        const v2f r2 = ld2c(at + 6);
        const v2f sx = ld2c(at) - ox, sy = ld2c(at + 2) - oy, sz = ld2c(at + 4) - oz;
        const v2f cn2 = (sx * sx + sy * sy) + sz * sz;
        const v2f proj = (dx * sx + dy * sy) + dz * sz;
        const v2f c2ray = cn2 - proj * proj;
        const v2f disc = r2 - c2ray;
        const float cut0 = sqrtf(disc.x), cut1 = sqrtf(disc.y);
        const float ta = proj.x + ((cn2.x > r2.x + 1e-4f) ? -cut0 : cut0), tb = proj.y + ((cn2.y > r2.y + 1e-4f) ? -cut1 : cut1);
        occ = occ || flat_blocks(c2ray.x < r2.x, ta, lim) || flat_blocks(c2ray.y < r2.y, tb, lim);
    }
    return occ;
}
// Closest hit of ONE ray per lane below `lim`, two records per packed instruction (FlatScene::pairs): flat_loop<false> for kernels that hold
// one path per lane (the shade kernels that trace their continuation ray in place, k_generate's camera rays: shade_stage.hpp "rays traced in
// place").  Records are visited in flat_loop's order with flat_loop's arithmetic per (ray, record), so distance and winner are the same
// bit for bit.  A pair that straddles the border between a plain section and its coplanar-group section is treated as a group pair: its
// plain record may then be remembered as a runner-up, which only sends the ray to the reference-order arithmetic a little more often.
// Returns the winner's record index (-1: none); t = its distance, runner = a near-tied runner-up from a coplanar group (-1: none).
APT_D int flat_closest1(const FlatScene& fl, f3 o, f3 d, float lim, float& t_out, int& runner) {
    cf_ptr at = (cf_ptr)fl.pairs;
    const v2f ox = sp2(o.x), oy = sp2(o.y), oz = sp2(o.z), dx = sp2(d.x), dy = sp2(d.y), dz = sp2(d.z);
    FlatBest b; b.t = lim; b.idx = -1; b.runner = -1;
    auto solve = [&](cf_ptr r, v2f& t, v2f& u, v2f& v) {      // as in flat_any1
        const v2f sx = ox - ld2c(r), sy = oy - ld2c(r + 2), sz = oz - ld2c(r + 4);
        const v2f ux = ld2c(r + 6), uy = ld2c(r + 8), uz = ld2c(r + 10);
        const v2f vx = ld2c(r + 12), vy = ld2c(r + 14), vz = ld2c(r + 16);
        const v2f tx = ld2c(r + 18), ty = ld2c(r + 20), tz = ld2c(r + 22);
        const v2f t_o = fma2(tx, sx, fma2(ty, sy, tz * sz));
        const v2f t_d = fma2(tx, dx, fma2(ty, dy, tz * dz));
        v2f inv; inv.x = __builtin_amdgcn_rcpf(t_d.x); inv.y = __builtin_amdgcn_rcpf(t_d.y);
        t = -t_o * inv;
        const v2f px = fma2(t, dx, sx), py = fma2(t, dy, sy), pz = fma2(t, dz, sz);
        u = fma2(ux, px, fma2(uy, py, uz * pz));
        v = fma2(vx, px, fma2(vy, py, vz * pz));
    };
    int idx = 0;                                             // wave-uniform index of the pair's first record
    {   // parallelograms
        const int n = fl.n_quads + fl.n_quads_tie, n_plain = fl.n_quads >> 1;      // pairs whose records are both plain
        int j = 0;
        for (; j < n_plain; j++, idx += 2, at += 24) {
            v2f t, u, v; solve(at, t, u, v);
            const v2f a = u - sp2(0.5f), c = v - sp2(0.5f);
            b = flat_candidate<false>(b, fmaxf(fabsf(a.x), fabsf(c.x)) <= 0.5f, t.x, idx);
            b = flat_candidate<false>(b, fmaxf(fabsf(a.y), fabsf(c.y)) <= 0.5f, t.y, idx + 1);
        }
        for (; 2 * j < n; j++, idx += 2, at += 24) {
            v2f t, u, v; solve(at, t, u, v);
            const v2f a = u - sp2(0.5f), c = v - sp2(0.5f);
            b = flat_candidate<true>(b, fmaxf(fabsf(a.x), fabsf(c.x)) <= 0.5f, t.x, idx);
            b = flat_candidate<true>(b, 2 * j + 1 < n && fmaxf(fabsf(a.y), fabsf(c.y)) <= 0.5f, t.y, idx + 1);
        }
        idx = n;
    }
    {   // convex quadrilaterals
        const int n = fl.n_gquads + fl.n_gquads_tie, n_plain = fl.n_gquads >> 1, first = idx;
        int j = 0;
        for (; j < n_plain; j++, idx += 2, at += 36) {
            v2f t, u, v; solve(at, t, u, v);
            const v2f e1 = fma2(ld2c(at + 24), u, fma2(ld2c(at + 26), v, ld2c(at + 28))), e2 = fma2(ld2c(at + 30), u, fma2(ld2c(at + 32), v, ld2c(at + 34)));
            b = flat_candidate<false>(b, fminf(fminf(u.x, v.x), fminf(e1.x, e2.x)) >= 0.f, t.x, idx);
            b = flat_candidate<false>(b, fminf(fminf(u.y, v.y), fminf(e1.y, e2.y)) >= 0.f, t.y, idx + 1);
        }
        for (; 2 * j < n; j++, idx += 2, at += 36) {
            v2f t, u, v; solve(at, t, u, v);
            const v2f e1 = fma2(ld2c(at + 24), u, fma2(ld2c(at + 26), v, ld2c(at + 28))), e2 = fma2(ld2c(at + 30), u, fma2(ld2c(at + 32), v, ld2c(at + 34)));
            b = flat_candidate<true>(b, fminf(fminf(u.x, v.x), fminf(e1.x, e2.x)) >= 0.f, t.x, idx);
            b = flat_candidate<true>(b, 2 * j + 1 < n && fminf(fminf(u.y, v.y), fminf(e1.y, e2.y)) >= 0.f, t.y, idx + 1);
        }
        idx = first + n;
    }
    {   // triangles
        const int n = fl.n_tris + fl.n_tris_tie, n_plain = fl.n_tris >> 1, first = idx;
        int j = 0;
        for (; j < n_plain; j++, idx += 2, at += 24) {
            v2f t, u, v; solve(at, t, u, v);
            const v2f w = (sp2(1.0f) - u) - v;
            b = flat_candidate<false>(b, fminf(fminf(u.x, v.x), w.x) >= 0.f, t.x, idx);
            b = flat_candidate<false>(b, fminf(fminf(u.y, v.y), w.y) >= 0.f, t.y, idx + 1);
        }
        for (; 2 * j < n; j++, idx += 2, at += 24) {
            v2f t, u, v; solve(at, t, u, v);
            const v2f w = (sp2(1.0f) - u) - v;
            b = flat_candidate<true>(b, fminf(fminf(u.x, v.x), w.x) >= 0.f, t.x, idx);
            b = flat_candidate<true>(b, 2 * j + 1 < n && fminf(fminf(u.y, v.y), w.y) >= 0.f, t.y, idx + 1);
        }
        idx = first + n;
    }
    for (int j = 0; 2 * j < fl.n_spheres; j++, idx += 2, at += 8) {      // spheres: flat_loop's test (the reference's, tracer_base.py:184-199) on two spheres
        const v2f r2 = ld2c(at + 6);
        const v2f sx = ld2c(at) - ox, sy = ld2c(at + 2) - oy, sz = ld2c(at + 4) - oz;
        const v2f cn2 = (sx * sx + sy * sy) + sz * sz;
        const v2f proj = (dx * sx + dy * sy) + dz * sz;
        const v2f c2ray = cn2 - proj * proj;
        const v2f disc = r2 - c2ray;
        const float cut0 = sqrtf(disc.x), cut1 = sqrtf(disc.y);
        const float ta = proj.x + ((cn2.x > r2.x + 1e-4f) ? -cut0 : cut0), tb = proj.y + ((cn2.y > r2.y + 1e-4f) ? -cut1 : cut1);
        b = flat_candidate<false>(b, c2ray.x < r2.x, ta, idx);
        b = flat_candidate<false>(b, 2 * j + 1 < fl.n_spheres && c2ray.y < r2.y, tb, idx + 1);
    }
    if ((fl.n_quads_tie | fl.n_gquads_tie | fl.n_tris_tie) != 0) {      // a runner-up only means something next to a winner from a coplanar group (flat_loop)
        const int a0 = fl.n_quads, a1 = a0 + fl.n_quads_tie, b0 = a1 + fl.n_gquads, b1 = b0 + fl.n_gquads_tie, c0 = b1 + fl.n_tris, c1 = c0 + fl.n_tris_tie;
        const bool tie = (b.idx >= a0 && b.idx < a1) || (b.idx >= b0 && b.idx < b1) || (b.idx >= c0 && b.idx < c1);
        b.runner = tie ? b.runner : -1;
    } else b.runner = -1;
    t_out = b.t; runner = b.runner;
    return b.idx;
}
#endif
