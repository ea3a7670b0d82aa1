// bvh_wide.cpp — collapses the binary SAH tree of bvh_build.cpp into the 8-wide, quantised tree the traversal kernels walk.
//
// Why 8-wide: an 8-wide node decides three binary levels with one fetch (a ray makes ~5 node steps where the binary tree made ~15).
// Why 64 bytes (round 6; the node was 80 = Ylitie, Karras, Laine's, HPG 2017): the walk is bound by the number of scattered 16-byte
// accesses a ray makes - the vector-memory pipe of a CU serves about one lane-access per cycle whatever the bytes (tools/probe_gather.hip,
// profiles/NOTES.md round 6) - so a node is as expensive as its 16-byte pieces: five then, FOUR now.  What had to shrink, and how:
//   * the node's corner: three floats -> three 16-bit coordinates on a GLOBAL grid (scene box / 65535 per axis, power-of-two step); the ray is
//     transformed into grid units once (traverse.hpp make_walk_ray), after which every plane of every node is an exact small integer
//     `corner + q * 2^e`;
//   * per-axis exponents: a byte each -> four bits each (the node's own step is the global step times 2^e, 0 <= e <= 9);
//   * per-child meta: a byte each -> two MASKS of a byte each (which slots are leaves, which are inner nodes).  A leaf child is ONE primitive
//     (the binary tree is built with single-primitive leaves: measured 1 / 2 / 3 per leaf level or worse in rounds 2 and 5), its index is
//     tri_base + the number of leaf slots below it; an inner child's index child_base + the number of inner slots below it;
//   * child_base / tri_base: 24 bits each (16.7 M nodes / primitives; apt_scene_create refuses more).
//
// Node = 16 dwords:
//   [0]     corner.x | corner.y << 16          (grid units)              [1] corner.z | leaf mask << 16 | inner mask << 24
//   [2]     child_base | ex << 24 | ey << 28                             [3] tri_base | ez << 24
//   [4..5]  lo.x[8]   [6..7] lo.y[8]   [8..9] lo.z[8]   [10..11] hi.x[8]   [12..13] hi.y[8]   [14..15] hi.z[8]      (bytes, one per child slot)
// child box = corner + q * 2^e per axis in grid units; lo is rounded down and hi up and the corner lies at or below the node's box, so
// the decoded box always contains the exact one (which is itself padded, bvh_build.cpp) and the traversal stays conservative: results
// depend on the per-primitive tests only.  A scene whose smallest features are below 1 / 65535 of its extent gets boxes no finer than
// the global grid - more candidate primitives per ray, the same hits.
// Slot assignment: slot s (bits x y z) should hold the child that lies towards +x/+y/+z where its bit is set, so that a ray can
// visit the hit children of a node front to back just by walking the slots in the order `slot XOR ray octant` (no sorting).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <thread>
#include <limits>
#include <memory>
#include <vector>

#include "bvh_build.hpp"

namespace apt {
namespace {

struct Kid { int32_t link; float lo[3], hi[3]; };

float half_area(const Kid& k) {
    const float d0 = k.hi[0] - k.lo[0], d1 = k.hi[1] - k.lo[1], d2 = k.hi[2] - k.lo[2];
    return d0 * d1 + d1 * d2 + d0 * d2;
}
int32_t as_link(float f) { int32_t v; std::memcpy(&v, &f, 4); return v; }
uint32_t as_bits(float f) { uint32_t v; std::memcpy(&v, &f, 4); return v; }

void kids_of(const BvhData& b, int node, Kid out[2]) {
    const float* nd = b.nodes.data() + 16 * (size_t)node;
    for (int c = 0; c < 2; c++) {
        for (int a = 0; a < 3; a++) { out[c].lo[a] = nd[6 * c + a]; out[c].hi[a] = nd[6 * c + 3 + a]; }
        out[c].link = as_link(nd[12 + c]);
    }
}
bool is_empty_leaf(const Kid& k) { return k.link < 0 && ((~k.link) & 15) == 0; }

// One 8-wide node: the gathered children, their slots, and what the emission needs.
struct Item { int node2, node8, depth; };
struct Work {                          // plain data, never value-initialised (a level of the big scenes holds 10^5 of these)
    int node2, node8, depth;
    Kid kids[8]; int n_kids;
    int kid_in[8];                     // slot -> child (or -1)
    float lo[3], hi[3]; int ex[3]; int qo[3];      // node box, per-axis exponent, corner on the global grid
    int n_inner, n_tris;               // inner children, primitives of the leaf children
    int child_base, tri_base;          // assigned by the level's prefix sum
};

// gather up to eight children (open the inner child of largest surface area until none is left or the node is full), frame, slots
void plan_node(const BvhData& bvh2, const WideFrame& gf, Work& w) {
    Kid* kids = w.kids;
    kids_of(bvh2, w.node2, kids);
    int n = 2;
    for (;;) {
        if (n >= 8) break;
        int best = -1; float best_a = -1.f;
        for (int i = 0; i < n; i++)
            if (kids[i].link >= 0) { const float a = half_area(kids[i]); if (a > best_a) { best_a = a; best = i; } }
        if (best < 0) break;
        Kid two[2]; kids_of(bvh2, kids[best].link, two);
        kids[best] = two[0]; kids[n++] = two[1];
    }
    { int m = 0; for (int i = 0; i < n; i++) if (!is_empty_leaf(kids[i])) kids[m++] = kids[i]; n = m; }
    w.n_kids = n;
    // ---- node box, quantisation frame
    float* lo = w.lo; float* hi = w.hi;
    for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::max(); hi[a] = -std::numeric_limits<float>::max(); }
    for (int i = 0; i < n; i++) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], kids[i].lo[a]); hi[a] = std::max(hi[a], kids[i].hi[a]); }
    if (n == 0) for (int a = 0; a < 3; a++) lo[a] = hi[a] = 0.f;
    for (int a = 0; a < 3; a++) {
        // corner: the grid point at or below the node's lower face; exponent: the smallest step 2^e (grid units) whose 255 cells reach the upper face
        const double g0 = (double)gf.gmin[a], gs = (double)gf.gstep[a];
        long qo = (long)std::floor(((double)lo[a] - g0) / gs);
        qo = std::min(std::max(qo, 0L), 65535L);
        while (qo > 0 && g0 + (double)qo * gs > (double)lo[a]) qo--;
        w.qo[a] = (int)qo;
        const double ext = ((double)hi[a] - g0) / gs - (double)qo;           // grid units
        int e = 0;
        while (e < 15 && ext > 255.0 * std::ldexp(1.0, e)) e++;
        w.ex[a] = e;
    }
    // ---- slot assignment: greedy over dot(child centre - node centre, direction of the slot)
    int slot_of[8];
    for (int s = 0; s < 8; s++) w.kid_in[s] = -1;
    float cost[8][8];
    for (int i = 0; i < n; i++) {
        slot_of[i] = -1;
        for (int s = 0; s < 8; s++) {
            float c = 0.f;
            for (int a = 0; a < 3; a++) {
                const float rel = 0.5f * (kids[i].lo[a] + kids[i].hi[a]) - 0.5f * (lo[a] + hi[a]);
                c += ((s >> (2 - a)) & 1) ? -rel : rel;        // bit set: the child towards + on that axis costs least
            }
            cost[i][s] = c;
        }
    }
    for (int round = 0; round < n; round++) {
        int bi = -1, bs = -1; float bc = std::numeric_limits<float>::max();
        for (int i = 0; i < n; i++) if (slot_of[i] < 0)
            for (int s = 0; s < 8; s++) if (w.kid_in[s] < 0 && cost[i][s] < bc) { bc = cost[i][s]; bi = i; bs = s; }
        slot_of[bi] = bs; w.kid_in[bs] = bi;
    }
    w.n_inner = 0; w.n_tris = 0;
    for (int s = 0; s < 8; s++) {
        if (w.kid_in[s] < 0) continue;
        const Kid& k = kids[w.kid_in[s]];
        if (k.link >= 0) w.n_inner++; else w.n_tris += (~k.link) & 15;      // (1: single-primitive leaves)
    }
}

// the 16 words of the node, its primitives' slots, its inner children's work items (in slot order, like the numbering)
int emit_node(const BvhData& bvh2, const WideFrame& gf, const Work& wk, uint32_t* nodes, int32_t* prim_order, Item* next) {
    uint32_t w[APT_NODE_DWORDS]; std::memset(w, 0, sizeof(w));
    uint8_t q[6][8];
    for (int s = 0; s < 8; s++) { for (int a = 0; a < 3; a++) { q[a][s] = 255; q[3 + a][s] = 0; } }      // empty slot: inverted box, never hit (and masked out by the two slot masks)
    uint32_t imask = 0, lmask = 0;
    int n_inner = 0, n_tris = 0;
    for (int s = 0; s < 8; s++) {
        if (wk.kid_in[s] < 0) continue;
        const Kid& k = wk.kids[wk.kid_in[s]];
        for (int a = 0; a < 3; a++) {
            const double g0 = (double)gf.gmin[a], gs = (double)gf.gstep[a], sc = std::ldexp(1.0, wk.ex[a]);
            const double klo = ((double)k.lo[a] - g0) / gs - (double)wk.qo[a], khi = ((double)k.hi[a] - g0) / gs - (double)wk.qo[a];      // grid units from the corner (exact: power-of-two step)
            long ql = (long)std::floor(klo / sc), qh = (long)std::ceil(khi / sc);
            ql = std::min(std::max(ql, 0L), 255L); qh = std::min(std::max(qh, 0L), 255L);
            while (ql > 0 && (double)ql * sc > klo) ql--;
            while (qh < 255 && (double)qh * sc < khi) qh++;
            if ((double)qh * sc < khi || (double)ql * sc > klo) return -2;              // frame too small: cannot happen (255 * 2^e >= extent, corner <= lower face)
            q[a][s] = (uint8_t)ql; q[3 + a][s] = (uint8_t)qh;
        }
        if (k.link >= 0) {
            imask |= 1u << s;
            Item& nx = next[n_inner];
            nx.node2 = k.link; nx.node8 = wk.child_base + n_inner; nx.depth = wk.depth + 1;
            n_inner++;
        } else {
            const int code = ~k.link, first = code >> 4, count = code & 15;
            if (count != 1) return -3;                                                  // the binary tree must be built with single-primitive leaves
            lmask |= 1u << s;
            prim_order[wk.tri_base + n_tris] = bvh2.prim_order[(size_t)first];
            n_tris++;
        }
    }
    if ((uint32_t)wk.child_base >= (1u << 24) || (uint32_t)wk.tri_base >= (1u << 24)) return -5;
    w[0] = (uint32_t)wk.qo[0] | ((uint32_t)wk.qo[1] << 16);
    w[1] = (uint32_t)wk.qo[2] | (lmask << 16) | (imask << 24);
    w[2] = (uint32_t)wk.child_base | ((uint32_t)wk.ex[0] << 24) | ((uint32_t)wk.ex[1] << 28);
    w[3] = (uint32_t)wk.tri_base | ((uint32_t)wk.ex[2] << 24);
    for (int g = 0; g < 6; g++) std::memcpy(&w[4 + 2 * g], q[g], 8);
    std::memcpy(nodes + APT_NODE_DWORDS * (size_t)wk.node8, w, sizeof(w));
    return 0;
}

}  // namespace

// Breadth first, one level at a time: the nodes of a level are planned in parallel, numbered by a prefix sum over the level (inner
// children -> node indices, leaf primitives -> slots, both in level order then slot order - exactly the numbering of a serial
// breadth-first walk), and emitted in parallel.  On the 1.14 M-triangle scene the collapse took 310 ms single-threaded, more than
// the device build of the binary tree (30 ms): the last levels hold almost all of the nodes, so they parallelise well.
int build_wide_bvh(const BvhData& bvh2, WideBvhData& out) {
    if (bvh2.n_nodes() <= 0) return -1;
    out.nodes.clear(); out.prim_order.clear(); out.max_depth = 0;
    out.prim_order.assign(bvh2.prim_order.size(), 0);
    const int threads = host_threads();
    // the global grid: the root's box (both children of binary node 0), 65535 cells of a power-of-two step per axis
    {
        Kid two[2]; kids_of(bvh2, 0, two);
        for (int a = 0; a < 3; a++) {
            float lo = std::numeric_limits<float>::max(), hi = -std::numeric_limits<float>::max();
            for (int c = 0; c < 2; c++) if (!is_empty_leaf(two[c])) { lo = std::min(lo, two[c].lo[a]); hi = std::max(hi, two[c].hi[a]); }
            if (!(lo <= hi)) { lo = 0.f; hi = 0.f; }
            const double ext = (double)hi - (double)lo;
            int e = (ext > 0.0) ? (int)std::ceil(std::log2(ext / 65534.0)) : -100;
            e = std::min(std::max(e, -100), 100);
            while (ext > 65534.0 * std::ldexp(1.0, e)) e++;
            out.frame.gmin[a] = lo; out.frame.gstep[a] = (float)std::ldexp(1.0, e);
        }
    }
    const WideFrame& gf = out.frame;
    std::vector<Item> level(1), next;
    level[0].node2 = 0; level[0].node8 = 0; level[0].depth = 1;
    std::unique_ptr<Work[]> work;
    size_t work_cap = 0;
    std::vector<int> next_at, rc;
    int n_nodes8 = 1, n_slots = 0;
    out.nodes.resize(APT_NODE_DWORDS, 0u);
    while (!level.empty()) {
        const int n = (int)level.size();
        out.max_depth = std::max(out.max_depth, level[0].depth);
        if ((size_t)n > work_cap) { work_cap = (size_t)n + (size_t)n / 2; work.reset(new Work[work_cap]); }
        Work* wk = work.get();
        parallel_for(n, threads, [&](int i) {
            Work& w = wk[i];
            w.node2 = level[(size_t)i].node2; w.node8 = level[(size_t)i].node8; w.depth = level[(size_t)i].depth;
            plan_node(bvh2, gf, w);
        });
        next_at.resize((size_t)n);
        const int first_child = n_nodes8;
        for (int i = 0; i < n; i++) {
            Work& w = wk[i];
            w.child_base = n_nodes8; w.tri_base = n_slots;
            next_at[(size_t)i] = n_nodes8 - first_child;
            n_nodes8 += w.n_inner; n_slots += w.n_tris;
        }
        if ((size_t)n_slots > out.prim_order.size()) return -4;
        out.nodes.resize((size_t)n_nodes8 * APT_NODE_DWORDS);
        next.resize((size_t)(n_nodes8 - first_child));
        rc.assign((size_t)n, 0);
        parallel_for(n, threads, [&](int i) {
            rc[(size_t)i] = emit_node(bvh2, gf, wk[i], out.nodes.data(), out.prim_order.data(), next.data() + next_at[(size_t)i]);
        });
        for (int i = 0; i < n; i++) if (rc[(size_t)i] != 0) return rc[(size_t)i];
        level.swap(next);
    }
    return ((size_t)n_slots == bvh2.prim_order.size()) ? 0 : -4;
}

}  // namespace apt
