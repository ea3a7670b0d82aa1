// bvh_wide.cpp — collapses the binary SAH tree of bvh_build.cpp into the 8-wide, quantised tree the traversal kernels walk.
//
// Why: on MI355X the walk over the binary tree is bound by the LATENCY of its dependent node fetches (one 64-byte record per
// level, ~15 levels per ray on the 95 k-triangle scene, a tree that does not fit the 4 MiB L2 of an XCD), not by arithmetic.
// An 8-wide node decides three binary levels with one fetch, and with child boxes quantised to 8 bits per plane relative to the
// node's own box it takes 80 bytes for up to eight children: a third of the dependent round trips and a fifth of the node bytes.
// (Layout after Ylitie, Karras, Laine, "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", HPG 2017.)
//
// Node = 20 dwords:
//   [0..2]  p = lower corner of the node box (float)            [3] ex | ey << 8 | ez << 16 | imask << 24
//   [4]     index of the first INNER child (inner children are consecutive, in slot order)
//   [5]     index of the first primitive of the node's LEAF children (consecutive, in slot order, at most 3 per leaf)
//   [6..7]  meta[8], one byte per child slot: 0 = empty; inner: 0x20 | (24 + slot); leaf of n primitives at offset k (from [5]):
//           unary(n) << 5 | k   (unary = 1, 3, 7)
//   [8..9]  lo.x[8]   [10..11] lo.y[8]   [12..13] lo.z[8]   [14..15] hi.x[8]   [16..17] hi.y[8]   [18..19] hi.z[8]      (bytes)
// child box = p + q * 2^e per axis; lo is rounded down and hi up, so the decoded box always contains the exact one (which is
// itself padded, bvh_build.cpp) and the traversal stays conservative: results depend on the per-primitive tests only.
// Slot assignment: slot s (bits x y z) should hold the child that lies towards +x/+y/+z where its bit is set, so that a ray can
// visit the hit children of a node front to back just by walking the slots in the order `slot XOR ray octant` (no sorting).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
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

}  // namespace

int build_wide_bvh(const BvhData& bvh2, WideBvhData& out) {
    if (bvh2.n_nodes() <= 0) return -1;
    out.nodes.clear(); out.prim_order.clear(); out.max_depth = 0;
    out.prim_order.reserve(bvh2.prim_order.size());
    struct Item { int node2, node8, depth; };
    std::deque<Item> queue;
    out.nodes.resize(20, 0u);
    queue.push_back({0, 0, 1});
    int n_nodes8 = 1;
    while (!queue.empty()) {
        const Item it = queue.front(); queue.pop_front();
        out.max_depth = std::max(out.max_depth, it.depth);
        // ---- gather up to eight children: open the inner child of largest surface area until none is left or the node is full
        std::vector<Kid> kids(2);
        kids_of(bvh2, it.node2, kids.data());
        for (;;) {
            if ((int)kids.size() >= 8) break;
            int best = -1; float best_a = -1.f;
            for (int i = 0; i < (int)kids.size(); i++)
                if (kids[(size_t)i].link >= 0) { const float a = half_area(kids[(size_t)i]); if (a > best_a) { best_a = a; best = i; } }
            if (best < 0) break;
            Kid two[2]; kids_of(bvh2, kids[(size_t)best].link, two);
            kids[(size_t)best] = two[0]; kids.push_back(two[1]);
        }
        kids.erase(std::remove_if(kids.begin(), kids.end(), is_empty_leaf), kids.end());
        // ---- node box, quantisation frame
        float lo[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()}, hi[3] = {-lo[0], -lo[0], -lo[0]};
        for (const Kid& k : kids) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], k.lo[a]); hi[a] = std::max(hi[a], k.hi[a]); }
        if (kids.empty()) for (int a = 0; a < 3; a++) lo[a] = hi[a] = 0.f;
        int ex[3];
        for (int a = 0; a < 3; a++) {
            const double ext = (double)hi[a] - (double)lo[a];
            int e = (ext > 0.0) ? (int)std::ceil(std::log2(ext / 255.0)) : -100;
            while (ext > 255.0 * std::ldexp(1.0, e)) e++;           // log2 rounding
            ex[a] = std::min(std::max(e, -100), 100);
        }
        // ---- slot assignment: greedy over dot(child centre - node centre, direction of the slot)
        int slot_of[8], kid_in[8];
        for (int s = 0; s < 8; s++) kid_in[s] = -1;
        {
            float cost[8][8];
            const int n = (int)kids.size();
            for (int i = 0; i < n; i++) {
                slot_of[i] = -1;
                for (int s = 0; s < 8; s++) {
                    float c = 0.f;
                    for (int a = 0; a < 3; a++) {
                        const float rel = 0.5f * (kids[(size_t)i].lo[a] + kids[(size_t)i].hi[a]) - 0.5f * (lo[a] + hi[a]);
                        c += ((s >> (2 - a)) & 1) ? -rel : rel;        // bit set: the child towards + on that axis costs least
                    }
                    cost[i][s] = c;
                }
            }
            for (int round = 0; round < n; round++) {
                int bi = -1, bs = -1; float bc = std::numeric_limits<float>::max();
                for (int i = 0; i < n; i++) if (slot_of[i] < 0)
                    for (int s = 0; s < 8; s++) if (kid_in[s] < 0 && cost[i][s] < bc) { bc = cost[i][s]; bi = i; bs = s; }
                slot_of[bi] = bs; kid_in[bs] = bi;
            }
        }
        // ---- emit
        uint32_t w[20]; std::memset(w, 0, sizeof(w));
        uint8_t meta[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[6][8];
        for (int s = 0; s < 8; s++) { for (int a = 0; a < 3; a++) { q[a][s] = 255; q[3 + a][s] = 0; } }      // empty slot: inverted box, never hit
        uint32_t imask = 0;
        int n_inner = 0, n_tris = 0;
        const int child_base = n_nodes8, tri_base = (int)out.prim_order.size();
        for (int s = 0; s < 8; s++) {
            if (kid_in[s] < 0) continue;
            const Kid& k = kids[(size_t)kid_in[s]];
            for (int a = 0; a < 3; a++) {
                const double sc = std::ldexp(1.0, ex[a]);
                long ql = (long)std::floor(((double)k.lo[a] - (double)lo[a]) / sc), qh = (long)std::ceil(((double)k.hi[a] - (double)lo[a]) / sc);
                ql = std::min(std::max(ql, 0L), 255L); qh = std::min(std::max(qh, 0L), 255L);
                while (ql > 0 && (double)lo[a] + (double)ql * sc > (double)k.lo[a]) ql--;
                while (qh < 255 && (double)lo[a] + (double)qh * sc < (double)k.hi[a]) qh++;
                if ((double)lo[a] + (double)qh * sc < (double)k.hi[a]) return -2;              // frame too small: cannot happen (255 * 2^e >= extent)
                q[a][s] = (uint8_t)ql; q[3 + a][s] = (uint8_t)qh;
            }
            if (k.link >= 0) {
                meta[s] = (uint8_t)(0x20 | (24 + s));
                imask |= 1u << s;
                queue.push_back({k.link, child_base + n_inner, it.depth + 1});
                n_inner++;
            } else {
                const int code = ~k.link, first = code >> 4, count = code & 15;
                if (count < 1 || count > 3 || n_tris + count > 24) return -3;                   // the binary tree must be built with max_leaf <= 3
                meta[s] = (uint8_t)((((1u << count) - 1u) << 5) | (uint32_t)n_tris);
                for (int k2 = 0; k2 < count; k2++) out.prim_order.push_back(bvh2.prim_order[(size_t)(first + k2)]);
                n_tris += count;
            }
        }
        n_nodes8 += n_inner;
        w[0] = as_bits(lo[0]); w[1] = as_bits(lo[1]); w[2] = as_bits(lo[2]);
        w[3] = ((uint32_t)(uint8_t)(int8_t)ex[0]) | ((uint32_t)(uint8_t)(int8_t)ex[1] << 8) | ((uint32_t)(uint8_t)(int8_t)ex[2] << 16) | (imask << 24);
        w[4] = (uint32_t)child_base; w[5] = (uint32_t)tri_base;
        std::memcpy(&w[6], meta, 8);
        for (int g = 0; g < 6; g++) std::memcpy(&w[8 + 2 * g], q[g], 8);
        if (out.nodes.size() < (size_t)n_nodes8 * 20) out.nodes.resize((size_t)n_nodes8 * 20, 0u);
        std::memcpy(out.nodes.data() + 20 * (size_t)it.node8, w, sizeof(w));
    }
    return (out.prim_order.size() == bvh2.prim_order.size()) ? 0 : -4;
}

}  // namespace apt
