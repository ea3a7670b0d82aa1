// traverse.hpp — software BVH traversal for gfx950 (no RT hardware on CDNA4).
//
// Replaces the reference's intersectors: brute force `TracerBase.ray_intersect/does_intersect`
// (tracer/tracer_base.py:168-278) and the stackless preorder walk
// `PathTracer.ray_intersect_bvh/does_intersect_bvh` (tracer/path_tracer.py:338-422).
// The per-primitive tests are the reference's, operation for operation (triangle: solve
// [e1 e2 -d] (u v t)^T = o - p0 with the adjugate inverse; sphere: tracer_base.py:184-199),
// so the closest hit is the same hit; the tree around them is ours (bvh_build.cpp):
//   * ordered traversal (near child first) with a per-lane stack kept in LDS, laid out
//     [depth][lane] so a wave's pushes/pops are bank-conflict free,
//   * the top `n_staged` nodes and (for small scenes) every primitive record are staged into
//     LDS once per workgroup; deeper nodes come from global memory through L1/L2.
#pragma once
#include "vec.hpp"

struct DevBvh {
    const float4* nodes;     // 4 float4 per node (bvh_build.cpp layout)
    const float4* prims;     // 3 float4 per primitive, BVH order
    int n_nodes, n_prims;
};
// primitive record: triangle q0=(p0, e1.x) q1=(e1.yz, e2.xy) q2=(e2.z, prim_id, 0, -)
//                   sphere   q0=(centre, r)                  q2=(-, prim_id, 1, -)

struct StagedBvh {           // LDS view; falls through to global beyond the staged prefix
    const float4* g_nodes; const float4* g_prims;
    const float4* s_nodes; const float4* s_prims;
    int n_staged_nodes, n_staged_prims;
    APT_D float4 node(int i, int q) const { return (i < n_staged_nodes) ? s_nodes[4 * i + q] : g_nodes[4 * i + q]; }
    APT_D float4 prim(int i, int q) const { return (i < n_staged_prims) ? s_prims[3 * i + q] : g_prims[3 * i + q]; }
};

// cooperative copy of the staged prefix; call from every thread of the block, then __syncthreads()
APT_D void stage_bvh(const DevBvh& b, float4* s_nodes, int cap_nodes, float4* s_prims, int cap_prims, StagedBvh& out) {
    int nn = min(b.n_nodes, cap_nodes), np = (b.n_prims <= cap_prims) ? b.n_prims : 0;
    for (int i = threadIdx.x; i < 4 * nn; i += blockDim.x) s_nodes[i] = b.nodes[i];
    for (int i = threadIdx.x; i < 3 * np; i += blockDim.x) s_prims[i] = b.prims[i];
    out.g_nodes = b.nodes; out.g_prims = b.prims; out.s_nodes = s_nodes; out.s_prims = s_prims;
    out.n_staged_nodes = nn; out.n_staged_prims = np;
}

struct HitRec { float t; int prim; float u, v; };

// entry distance of the slab test, or -1 when the box is missed / behind / beyond tmax
APT_D float box_entry(f3 lo, f3 hi, f3 o, f3 inv_d, float tmax) {
    f3 t0 = (lo - o) * inv_d, t1 = (hi - o) * inv_d;
    float tn = max3(min3v(t0, t1));
    float tf = min3(max3v(t0, t1));
    return (tn <= tf && tf > 0.f && tn <= tmax) ? fmaxf(tn, 0.f) : -1.f;
}

// One primitive against the ray.  Returns the reference's ray_t (or -1) and barycentrics.
APT_D float prim_test(float4 q0, float4 q1, float4 q2, f3 o, f3 d, float& u, float& v) {
    u = 0.f; v = 0.f;
    if (__float_as_int(q2.z) != 0) {          // sphere: tracer_base.py:184-199
        f3 c = mk3(q0.x, q0.y, q0.z);
        float r2 = q0.w * q0.w;
        f3 s2c = c - o;
        float cn2 = norm2(s2c);
        float proj = dot(d, s2c);
        float c2ray = cn2 - proj * proj;
        if (c2ray >= r2) return -1.f;
        float cut = sqrtf(r2 - c2ray);
        return proj + ((cn2 > r2 + 1e-4f) ? -cut : cut);
    }
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

// ANY = false: closest hit, rec.t starts at the search limit and ends at min_depth.
// ANY = true : returns true on the first hit with 1e-4 < t < rec.t.
// `stack` points at this lane's column of an LDS array [depth][stride].
//
// Loop shape ("while-while"): all lanes of the wave walk inner nodes together until each of them holds a
// leaf (or is finished), then all of them test primitives together.  One mixed loop, where some lanes do a
// box pair while others do up to four primitive tests, costs the sum of both bodies on every iteration.
#define APT_TRAV_DONE ((int)0x80000000)      // not a valid leaf link: ~link would be first_prim = 2^27
template <bool ANY>
APT_D bool traverse(const StagedBvh& bvh, int* stack, int stride, f3 o, f3 d, HitRec& rec) {
    const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    int sp = 0;
    int cur = 0;
    while (cur != APT_TRAV_DONE) {
        while (cur >= 0) {
            float4 q0 = bvh.node(cur, 0), q1 = bvh.node(cur, 1), q2 = bvh.node(cur, 2), q3 = bvh.node(cur, 3);
            float tl = box_entry(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), o, inv_d, rec.t);
            float tr = box_entry(mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), o, inv_d, rec.t);
            int l = __float_as_int(q3.x), r = __float_as_int(q3.y);
            bool hl = tl >= 0.f, hr = tr >= 0.f;
            if (hl && hr) {
                bool swap = tr < tl;
                stack[sp * stride] = swap ? l : r; sp++;          // far child waits
                cur = swap ? r : l;
            } else if (hl) cur = l;
            else if (hr) cur = r;
            else if (sp > 0) { sp--; cur = stack[sp * stride]; }
            else cur = APT_TRAV_DONE;
        }
        while (cur < 0 && cur != APT_TRAV_DONE) {
            int code = ~cur;
            int first = code >> 4, count = code & 15;
            for (int k = 0; k < count; k++) {
                float4 p0 = bvh.prim(first + k, 0), p1 = bvh.prim(first + k, 1), p2 = bvh.prim(first + k, 2);
                float u, v;
                float t = prim_test(p0, p1, p2, o, d, u, v);
                if (t > 1e-4f && t < rec.t) {
                    if (ANY) return true;
                    rec.t = t; rec.prim = __float_as_int(p2.y); rec.u = u; rec.v = v;
                }
            }
            if (sp > 0) { sp--; cur = stack[sp * stride]; }
            else cur = APT_TRAV_DONE;
        }
    }
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
// 16-byte aligned record quarter -> float4 (four adjacent scalar dwords; merged into one s_load_dwordx4)
APT_D float4 ld4c(cf_ptr p) { return make_float4(p[0], p[1], p[2], p[3]); }
struct SweepScene {
    const float4* recs;       // 4 float4 per primitive, ORIGINAL order:
                              //   triangle (p0, e1.x) (e1.yz, e2.xy) (e2.z, n.xyz) with n = t-row cofactors of [e1 e2 .]
                              //   sphere   (centre, r)
    const float* obj_aabb;    // n_objects * 6
    const int* obj_info;      // n_objects * 3: first prim, count, is_sphere
    int n_objects;
};

template <bool ANY>
APT_D bool sweep(const SweepScene& sc, f3 o, f3 d, HitRec& rec) {
    bool found = false;                                   // ANY: this lane already has its answer
    const cf_ptr recs = (cf_ptr)(const float*)sc.recs;
    const cf_ptr aabb = (cf_ptr)sc.obj_aabb;
    const ci_ptr info = (ci_ptr)sc.obj_info;
    const f3 inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    for (int ob = 0; ob < sc.n_objects; ob++) {
        const int first = info[3 * ob], count = info[3 * ob + 1];
        // Per-object slab cull (TracerBase.aabb_test, tracer_base.py:159-166 and the `t_near > min_depth`
        // skip at :184).  It is part of the result, not just a speed-up: a ray with a zero direction
        // component whose origin sits on (or a rounding error outside) a slab boundary is culled upstream
        // even though the triangle test would accept it.  The reference divides by the ray; dividing six
        // times per object costs more than testing a triangle, so a reciprocal-multiply version with an
        // error band decides the clear cases and only lanes inside the band (or with non-finite slabs)
        // take the division path.
        bool need = !found;
        {
            const cf_ptr bb = aabb + 6 * ob;
            const f3 lo = mk3(bb[0], bb[1], bb[2]) - o, hi = mk3(bb[3], bb[4], bb[5]) - o;
            f3 t0 = lo * inv_d, t1 = hi * inv_d;
            float tn = max3(min3v(t0, t1)), tf = min3(max3v(t0, t1));
            const float band = 2e-6f * (fabsf(tn) + fabsf(tf));
            const bool pass = (tn + band < tf) && (tf > 1e-30f) && (tn + band < rec.t);
            const bool fail = (tn > tf + band) || (tf < 0.f) || (tn > rec.t + band);
            bool ok = pass;
            if (!(pass || fail)) {
                t0 = mk3(lo.x / d.x, lo.y / d.y, lo.z / d.z); t1 = mk3(hi.x / d.x, hi.y / d.y, hi.z / d.z);
                tn = max3(min3v(t0, t1)); tf = min3(max3v(t0, t1));
                ok = (tn < tf) && tf > 0.f && !(tn > rec.t);
            }
            need = need && ok;
        }
        if (!__any(need)) continue;                       // wave-uniform skip
        if (info[3 * ob + 2]) {
            float4 q0 = ld4c(recs + 16 * first);
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
        } else {
            for (int k = first; k < first + count; k++) {
                float4 q0 = ld4c(recs + 16 * k), q1 = ld4c(recs + 16 * k + 4), q2 = ld4c(recs + 16 * k + 8);
                float a00 = q0.w, a10 = q1.x, a20 = q1.y;      // e1
                float a01 = q1.z, a11 = q1.w, a21 = q2.x;      // e2
                float a02 = -d.x, a12 = -d.y, a22 = -d.z;
                float c00 = a11 * a22 - a21 * a12, c01 = a21 * a02 - a01 * a22, c02 = a01 * a12 - a11 * a02;
                float det = (a00 * c00 + a10 * c01) + a20 * c02;
                float inv_det = 1.0f / det;
                f3 s = o - mk3(q0.x, q0.y, q0.z);
                float c10 = a12 * a20 - a22 * a10, c11 = a22 * a00 - a02 * a20, c12 = a02 * a10 - a12 * a00;
                float u = ((inv_det * c00) * s.x + (inv_det * c01) * s.y) + (inv_det * c02) * s.z;
                float v = ((inv_det * c10) * s.x + (inv_det * c11) * s.y) + (inv_det * c12) * s.z;
                float t = ((inv_det * q2.y) * s.x + (inv_det * q2.z) * s.y) + (inv_det * q2.w) * s.z;
                if (need && u >= 0.f && v >= 0.f && u + v <= 1.0f && t > 1e-4f && t < rec.t) {
                    if (ANY) { found = true; need = false; }
                    else { rec.t = t; rec.prim = k; rec.u = u; rec.v = v; }
                }
            }
        }
        if (ANY && __all(found)) break;
    }
    return found;
}
APT_D bool sweep_any(const SweepScene& sc, f3 o, f3 d, HitRec& rec) { return sweep<true>(sc, o, d, rec); }
