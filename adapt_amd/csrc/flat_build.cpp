// flat_build.cpp — host side of the flat sweep (traverse.hpp "Flat sweep"): the scene's primitives as precomputed-transform records.
//
// Replaces, for small scenes in the fast build, the data the reference's brute-force intersector reads per ray: `prims` / `precom_vec`
// (tracer/tracer_base.py:117-134, 184-212).  Per planar primitive its corner p0 and the rows U, V, T of [e1 e2 n]^-1 (n = e1 x e2), so
// that for a point x:  u = U . (x - p0), v = V . (x - p0), height over the plane = T . (x - p0)
// (Baldwin & Weber, "Fast Ray-Triangle Intersections by Coordinate Transformation", JCGT 5(3), 2016).  Computed in double, stored as
// float.  Two coplanar triangles of one object that share an edge and form a convex outline become ONE record in the basis (corner opposite
// the shared edge, edge, edge): a parallelogram (inside <=> u, v in [0, 1]) or a general convex quadrilateral (u, v >= 0 and two more edge
// functions of (u, v)); which triangle a hit belongs to is u + v <= 1, and each triangle's own barycentrics are an affine map of the
// record's (u, v) (coefficients in {-1, 0, 1} for a parallelogram).  Degenerate triangles get a record no ray can hit (upstream: det = 0 -> non-finite barycentrics -> never accepted).
#include <cmath>
#include <cstring>

#include "bvh_build.hpp"

namespace apt {
namespace {
struct D3 { double x, y, z; };
inline D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 add(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline D3 vtx(const float* prims, int k, int v) { const float* p = prims + 9 * (size_t)k + 3 * v; return {p[0], p[1], p[2]}; }

struct Planar { float P0[3], U[4], V[4], T[4]; int prim_a, prim_b; float map_a[6], map_b[6]; bool ok; int obj; bool tie; float lo[3], hi[3]; float far_edges[6]; };

// rows of [e1 e2 n]^-1 applied to (x - p0)
bool make_rows(D3 p0, D3 e1, D3 e2, Planar& r) {
    const D3 n = cross(e1, e2);
    const double det = dot(n, n);
    if (!(det > 0.0) || !std::isfinite(det)) return false;
    const D3 ru = cross(e2, n), rv = cross(n, e1);
    const D3 U = {ru.x / det, ru.y / det, ru.z / det}, V = {rv.x / det, rv.y / det, rv.z / det};
    // t = -T(s) / T(d) does not depend on the scale or sign of the T row: store the plane in a canonical form - unit normal with its first
    // non-zero component positive, offset rounded once - so that COPLANAR primitives (a box resting on the floor, a decal on a wall) are
    // recognisable by bit-identical rows (T[3], the offset, serves only that comparison).  Their records go to the tie sections of the
    // stream, where the reference's own arithmetic settles which of two faces at the same distance is "hit" (traverse.hpp flat_tie_break).
    const double len = std::sqrt(det);
    D3 T = {n.x / len, n.y / len, n.z / len};
    auto snap = [](double a) { return std::fabs(a) < 1e-12 ? 0.0 : ((std::fabs(std::fabs(a) - 1.0) < 1e-12) ? std::copysign(1.0, a) : a); };
    T = {snap(T.x), snap(T.y), snap(T.z)};
    const double lead = (T.x != 0.0) ? T.x : ((T.y != 0.0) ? T.y : T.z);
    if (lead < 0.0) T = {-T.x, -T.y, -T.z};
    T = {T.x + 0.0, T.y + 0.0, T.z + 0.0};                                    // -0 -> +0: the rows are compared bit for bit
    T = {(double)(float)T.x, (double)(float)T.y, (double)(float)T.z};        // the stored normal; the offset below belongs to exactly these floats
    const double uw = -dot(U, p0), vw = -dot(V, p0), tw = -dot(T, p0) + 0.0;
    const double all[12] = {U.x, U.y, U.z, uw, V.x, V.y, V.z, vw, T.x, T.y, T.z, tw};
    for (double a : all) if (!std::isfinite(a) || std::fabs(a) > 1e30) return false;
    r.P0[0] = (float)p0.x; r.P0[1] = (float)p0.y; r.P0[2] = (float)p0.z;      // exact: p0 is one of the float32 vertices
    r.U[0] = (float)U.x; r.U[1] = (float)U.y; r.U[2] = (float)U.z; r.U[3] = (float)uw;
    r.V[0] = (float)V.x; r.V[1] = (float)V.y; r.V[2] = (float)V.z; r.V[3] = (float)vw;
    r.T[0] = (float)T.x; r.T[1] = (float)T.y; r.T[2] = (float)T.z; r.T[3] = (float)tw;
    return true;
}
// triangle k's own barycentrics as an affine map of the record's (u, v) - k lies in the record's plane, so the map is exact: evaluate
// k's barycentrics at the record's corner and edge ends.  (For a parallelogram half the coefficients come out in {-1, 0, 1}.)
bool bary_map(const float* prims, int k, D3 p0, D3 e1, D3 e2, float m[6]) {
    const D3 q0 = vtx(prims, k, 0), f1 = sub(vtx(prims, k, 1), q0), f2 = sub(vtx(prims, k, 2), q0);
    const D3 n = cross(f1, f2); const double det = dot(n, n);
    if (!(det > 0.0) || !std::isfinite(det)) return false;
    auto bary = [&](D3 x, double& a, double& b) { const D3 w = sub(x, q0); a = dot(cross(w, f2), n) / det; b = dot(cross(f1, w), n) / det; };
    double a0, b0, a1, b1, a2, b2;
    bary(p0, a0, b0); bary(add(p0, e1), a1, b1); bary(add(p0, e2), a2, b2);
    const double c[6] = {a0, a1 - a0, a2 - a0, b0, b1 - b0, b2 - b0};
    for (int i = 0; i < 6; i++) {
        const double r = std::nearbyint(c[i]);
        m[i] = (float)((std::fabs(c[i] - r) < 1e-9) ? r + 0.0 : c[i]);
    }
    return true;
}

// Do two convex polygons of one plane overlap with positive area?  Separating-axis test over the in-plane normals of both polygons'
// edges, with a margin: polygons that only share an edge or a corner (the two triangles of a non-parallelogram face) do not count.
typedef std::vector<D3> Poly;
bool convex_overlap(const Poly& A, const Poly& B, const float plane[4]) {
    const D3 n = {plane[0], plane[1], plane[2]};
    double ext = 0.0;
    for (const D3& p : A) for (const D3& q : A) ext = std::max(ext, std::sqrt(dot(sub(p, q), sub(p, q))));
    const double margin = 1e-5 * ext + 1e-9;
    for (int pass = 0; pass < 2; pass++) {
        const Poly& P = pass ? B : A;
        for (size_t i = 0; i < P.size(); i++) {
            const D3 e = sub(P[(i + 1) % P.size()], P[i]);
            D3 ax = cross(n, e);
            const double len = std::sqrt(dot(ax, ax));
            if (len < 1e-20) continue;
            ax = {ax.x / len, ax.y / len, ax.z / len};
            double a0 = 1e300, a1 = -1e300, b0 = 1e300, b1 = -1e300;
            for (const D3& p : A) { const double t = dot(ax, p); a0 = std::min(a0, t); a1 = std::max(a1, t); }
            for (const D3& p : B) { const double t = dot(ax, p); b0 = std::min(b0, t); b1 = std::max(b1, t); }
            if (std::min(a1, b1) - std::max(a0, b0) <= margin) return false;       // a separating axis (or mere contact)
        }
    }
    return true;
}
}  // namespace

// One triangle as the BVH walk of the product build reads it (traverse.hpp tri_two): corner p0, rows U, V, T of [e1 e2 n]^-1, 12 floats.
// A degenerate triangle gets all-zero rows: T . d = 0 -> t is NaN or inf -> never accepted (upstream: det = 0 -> NaN barycentrics).
void planar_rows(const float* tri9, float out12[12]) {
    const D3 p0 = {tri9[0], tri9[1], tri9[2]}, e1 = sub({tri9[3], tri9[4], tri9[5]}, p0), e2 = sub({tri9[6], tri9[7], tri9[8]}, p0);
    Planar r;
    for (int a = 0; a < 12; a++) out12[a] = 0.f;
    out12[0] = tri9[0]; out12[1] = tri9[1]; out12[2] = tri9[2];
    if (!make_rows(p0, e1, e2, r)) return;
    for (int a = 0; a < 3; a++) { out12[3 + a] = r.U[a]; out12[6 + a] = r.V[a]; out12[9 + a] = r.T[a]; }
}

// stream: [parallelograms][same, of coplanar groups] x 12 floats, [convex quads][same, of coplanar groups] x 18, [triangles][same, of
// coplanar groups] x 12, [spheres] x 4; a planar record = corner p0, rows U, V, T (3 floats each); a convex quad appends its two far
// edges as functions a u + b v + c of the record's (u, v) that are >= 0 inside.  tab: 28 floats per record, in stream order (7 float4: (U, p0.x), (V, p0.y), (prim_a, prim_b, class_a,
// class_b), map_a, map_b, (p0.z, -, -, -)).
// prim_class: material class per primitive (sorted shading) or null.  transmissive: per object, 1 when rays can travel inside it (a BSDF).
// A record joins a coplanar group when another record lies in the same plane (to 2e-5) and their outlines overlap - the configurations in
// which upstream's answer hangs on the last bit of two distances (traverse.hpp flat_tie_break): a glass box resting on the floor, a decal on a wall.
int build_flat(const float* prims, int n_prims, const int32_t* obj_info, int n_objects, const int32_t* prim_class, const uint8_t* transmissive,
               std::vector<float>& stream, std::vector<float>& tab, int counts[7]) {
    std::vector<Planar> quads, gquads, tris;
    std::vector<int> spheres;
    std::vector<uint8_t> used((size_t)n_prims, 0);
    const float ident[6] = {0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    auto bounds = [&](Planar& p) {
        for (int a = 0; a < 3; a++) { p.lo[a] = 1e30f; p.hi[a] = -1e30f; }
        for (int k : {p.prim_a, p.prim_b}) if (k >= 0) for (int v = 0; v < 3; v++) for (int a = 0; a < 3; a++) {
            const float x = prims[9 * (size_t)k + 3 * v + a]; p.lo[a] = std::min(p.lo[a], x); p.hi[a] = std::max(p.hi[a], x);
        }
    };
    for (int o = 0; o < n_objects; o++) {
        const int first = obj_info[3 * o], count = obj_info[3 * o + 1];
        if (first < 0 || count < 0 || first + count > n_prims) return -1;
        if (obj_info[3 * o + 2]) { for (int k = first; k < first + count; k++) spheres.push_back(k); continue; }
        for (int k = first; k < first + count; k++) {
            if (used[(size_t)k]) continue;
            used[(size_t)k] = 1;
            const D3 A = vtx(prims, k, 0), B = vtx(prims, k, 1), C = vtx(prims, k, 2);
            const double ext = std::sqrt(std::max(dot(sub(B, A), sub(B, A)), dot(sub(C, A), sub(C, A)))), tol = 1e-6 * ext + 1e-12;
            // a partner in the same object that shares an edge, lies in the same plane and makes a convex outline with this triangle.  In this
            // triangle's frame (corner = the vertex opposite the shared edge) the partner's third vertex sits at (ud, vd): (1, 1) closes a
            // parallelogram; any other point with ud > 0, vd > 0, ud + vd > 1 a convex quadrilateral.
            bool paired = false;
            const D3 tv[3] = {A, B, C};
            for (int k2 = k + 1; k2 < first + count && !paired; k2++) {
                if (used[(size_t)k2]) continue;
                const D3 w[3] = {vtx(prims, k2, 0), vtx(prims, k2, 1), vtx(prims, k2, 2)};
                for (int opp = 0; opp < 3 && !paired; opp++) {            // this triangle's vertex opposite the shared edge
                    const D3 s1 = tv[(opp + 1) % 3], s2 = tv[(opp + 2) % 3];
                    auto same = [&](D3 a, D3 b) { const D3 d = sub(a, b); return dot(d, d) <= tol * tol; };
                    int hit_s1 = -1, hit_s2 = -1;
                    for (int j = 0; j < 3; j++) { if (same(w[j], s1)) hit_s1 = j; else if (same(w[j], s2)) hit_s2 = j; }
                    if (hit_s1 < 0 || hit_s2 < 0 || hit_s1 == hit_s2) continue;
                    const D3 far = w[3 - hit_s1 - hit_s2];
                    const D3 p0 = tv[opp], e1 = sub(s1, p0), e2 = sub(s2, p0), n = cross(e1, e2);
                    const double det = dot(n, n);
                    if (!(det > 0.0)) continue;
                    const D3 fd = sub(far, p0);
                    if (std::fabs(dot(fd, n)) / std::sqrt(det) > tol) continue;                                  // not in this triangle's plane
                    const double ud = dot(cross(fd, e2), n) / det, vd = dot(cross(e1, fd), n) / det;
                    if (!(ud > 1e-4 && vd > 1e-4 && ud + vd > 1.0 + 1e-4)) continue;                             // the outline would not be convex
                    Planar q{}; q.prim_a = k; q.prim_b = k2; q.obj = o;
                    if (!make_rows(p0, e1, e2, q)) continue;
                    if (!bary_map(prims, k, p0, e1, e2, q.map_a) || !bary_map(prims, k2, p0, e1, e2, q.map_b)) continue;
                    q.ok = true; bounds(q); used[(size_t)k2] = 1; paired = true;
                    if (std::fabs(ud - 1.0) <= 1e-6 && std::fabs(vd - 1.0) <= 1e-6) { quads.push_back(q); break; }
                    // the far edges (1, 0) -> (ud, vd) and (ud, vd) -> (0, 1) as functions that are >= 0 inside
                    const double fe[6] = {-vd, ud - 1.0, vd, vd - 1.0, -ud, ud};
                    for (int i = 0; i < 6; i++) q.far_edges[i] = (float)fe[i];
                    gquads.push_back(q);
                }
            }
            if (paired) continue;
            Planar t{}; t.prim_a = k; t.prim_b = -1; t.obj = o; memcpy(t.map_a, ident, sizeof(ident)); memcpy(t.map_b, ident, sizeof(ident));
            t.ok = make_rows(A, sub(B, A), sub(C, A), t);
            bounds(t);
            tris.push_back(t);
        }
    }
    auto poly = [&](const Planar& p) {                      // the record's outline: triangle, or parallelogram corner, corner + e1, far corner, corner + e2
        Poly out;
        if (p.prim_b < 0) { for (int v = 0; v < 3; v++) out.push_back(vtx(prims, p.prim_a, v)); return out; }
        // the four distinct vertices of the two triangles, ordered around the centroid
        std::vector<D3> pts;
        for (int k : {p.prim_a, p.prim_b}) for (int v = 0; v < 3; v++) {
            const D3 x = vtx(prims, k, v); bool dup = false;
            for (const D3& y : pts) { const D3 dd = sub(x, y); if (dot(dd, dd) < 1e-12 * (1.0 + dot(x, x))) dup = true; }
            if (!dup) pts.push_back(x);
        }
        D3 c = {0, 0, 0}; for (const D3& x : pts) c = add(c, x); c = {c.x / pts.size(), c.y / pts.size(), c.z / pts.size()};
        const D3 n = {p.T[0], p.T[1], p.T[2]}, ref = sub(pts[0], c);
        std::vector<std::pair<double, int>> ang;
        for (size_t i = 0; i < pts.size(); i++) { const D3 r = sub(pts[i], c); ang.push_back({std::atan2(dot(cross(ref, r), n), dot(ref, r)), (int)i}); }
        std::sort(ang.begin(), ang.end());
        for (auto& a : ang) out.push_back(pts[(size_t)a.second]);
        return out;
    };
    {   // coplanar groups
        std::vector<Planar*> all;
        for (Planar& p : quads) all.push_back(&p);
        for (Planar& p : gquads) all.push_back(&p);
        for (Planar& p : tris) all.push_back(&p);
        for (size_t i = 0; i < all.size(); i++) for (size_t j = i + 1; j < all.size(); j++) {
            Planar &a = *all[i], &b = *all[j];
            if (!a.ok || !b.ok) continue;
            // the same plane up to what a ray can tell apart: the tie sections accept candidates within 1e-5 t + 1e-6 of each other, so planes
            // a few 1e-5 apart (a box "on" the floor whose transformed vertices sit at y = 1e-16) have to be in them
            bool same_plane = std::fabs((double)a.T[3] - (double)b.T[3]) <= 2e-5 * (1.0 + std::fabs((double)a.T[3]));
            for (int c = 0; c < 3; c++) if (std::fabs((double)a.T[c] - (double)b.T[c]) > 1e-5) same_plane = false;
            if (!same_plane) continue;
            (void)transmissive;      // (opaque pairs too: a decal on a wall is coplanar geometry seen from outside, and upstream's picture of it is decided by the same last bits)
            bool overlap = convex_overlap(poly(a), poly(b), a.T);
            if (overlap) { a.tie = true; b.tie = true; }
        }
    }
    auto split = [](std::vector<Planar>& v) {               // stable: plain records first, coplanar-group records after, scene order inside each
        std::vector<Planar> plain, tie;
        for (const Planar& p : v) (p.tie ? tie : plain).push_back(p);
        const int n_tie = (int)tie.size();
        v = plain; v.insert(v.end(), tie.begin(), tie.end());
        return n_tie;
    };
    const int nq_tie = split(quads), ng_tie = split(gquads), nt_tie = split(tris);
    counts[0] = (int)quads.size() - nq_tie; counts[1] = nq_tie; counts[2] = (int)gquads.size() - ng_tie; counts[3] = ng_tie;
    counts[4] = (int)tris.size() - nt_tie; counts[5] = nt_tie; counts[6] = (int)spheres.size();
    const int n_quads = (int)quads.size(), n_gquads = (int)gquads.size(), n_tris = (int)tris.size(), n_spheres = (int)spheres.size();
    stream.assign((size_t)(n_quads + n_tris) * 12 + (size_t)n_gquads * 18 + (size_t)n_spheres * 4 + 4, 0.f);
    tab.assign((size_t)(n_quads + n_gquads + n_tris + n_spheres) * 28, 0.f);
    size_t at = 0, rec = 0;
    auto cls_of = [&](int k) -> int32_t { return (prim_class && k >= 0) ? prim_class[k] : -1; };
    auto put_planar = [&](const std::vector<Planar>& v, int stride) {
        for (const Planar& p : v) {
            float* r = stream.data() + at;
            float* e = tab.data() + 28 * rec;
            int32_t ids[4] = {p.prim_a, p.ok ? p.prim_b : -1, cls_of(p.prim_a), cls_of(p.prim_b)};
            if (p.ok) {
                for (int c = 0; c < 3; c++) { r[c] = p.P0[c]; r[3 + c] = p.U[c]; r[6 + c] = p.V[c]; r[9 + c] = p.T[c]; e[c] = p.U[c]; e[4 + c] = p.V[c]; }
                e[3] = p.P0[0]; e[7] = p.P0[1]; e[24] = p.P0[2];
                memcpy(e + 12, p.map_a, 24); memcpy(e + 18, p.map_b, 24);
                if (stride == 18) memcpy(r + 12, p.far_edges, 24);
            }                                                    // a degenerate triangle keeps its slot with all-zero rows: t = -0 / 0 = NaN, which no comparison accepts
            memcpy(e + 8, ids, 16);
            at += (size_t)stride; rec++;
        }
    };
    put_planar(quads, 12);
    put_planar(gquads, 18);
    put_planar(tris, 12);
    for (int k : spheres) {
        float* r = stream.data() + at;
        float* e = tab.data() + 28 * rec;
        const float* p = prims + 9 * (size_t)k;
        r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3] * p[3];               // radius2 = r ** 2 in float32 (tracer_base.py:186)
        int32_t ids[4] = {k, -1, cls_of(k), -1};
        memcpy(e + 8, ids, 16);
        memcpy(e + 12, ident, 24); memcpy(e + 18, ident, 24);
        at += 4; rec++;
    }
    return 0;
}
}  // namespace apt
