// bvh_build.cpp — host-side BVH construction for the gfx950 traversal kernels.
//
// Replaces the reference's native builder `bvh_cpp.bvh_build` (tracer/bvh/bvh.cpp:274-296:
// recursive binned-SAH BVH over triangle + sphere AABBs, preorder-linearised with subtree
// skip offsets for a stackless, unordered walk).  The layout here is different on purpose:
// a binary BVH whose nodes hold BOTH children's boxes (one 64-byte fetch decides two
// subtrees and gives the near/far order), primitives re-ordered into leaf order, leaves of
// up to APT_MAX_LEAF primitives.  Closest-hit results do not depend on the tree (the
// per-primitive tests are the reference's), only the amount of work does.
//
// Node = 16 dwords: [0..5] left min/max, [6..11] right min/max, [12] left link, [13] right link.
// link >= 0: inner node index; link < 0: leaf, ~link = (first_prim << 4) | prim_count.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <thread>
#include <vector>

#include "bvh_build.hpp"

namespace apt {

namespace {
constexpr int kBins = 16;
constexpr float kTraverseCost = 1.0f;   // relative to one primitive test

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; } }
    void grow(const Box& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    void grow(const float p[3]) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    float half_area() const {
        float d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
        if (d0 < 0.f || d1 < 0.f || d2 < 0.f) return 0.f;
        return d0 * d1 + d1 * d2 + d0 * d2;
    }
};
struct Ref { Box box; float c[3]; int prim; };

// a subtree left for a worker thread: its primitive range, and the link slot of the node that waits for its root
struct Task { int first, count; Box box; int depth, parent, slot; };

struct Builder {
    Ref* refs = nullptr;          // shared array; a builder only touches the range it was given
    std::vector<float> nodes;     // 16 floats per node
    int max_depth = 0;
    int kMaxLeaf = 4;
    int grain = 0;                // > 0: subtrees of at most this many primitives are not built but recorded in `tasks`
    std::vector<Task>* tasks = nullptr;

    static float as_float(int32_t v) { float f; std::memcpy(&f, &v, 4); return f; }

    int32_t leaf_link(int first, int count) { return ~((first << 4) | count); }

    // returns link for the subtree over refs[first, first+count); `box` = its bounds
    int32_t build(int first, int count, const Box& box, int depth, int parent = -1, int slot = 0) {
        max_depth = std::max(max_depth, depth);
        if (count <= 1) return leaf_link(first, count);
        if (tasks && parent >= 0 && count <= grain) { tasks->push_back({first, count, box, depth, parent, slot}); return 0; }      // link patched in later
        Box cb; cb.reset();
        for (int i = first; i < first + count; i++) cb.grow(refs[i].c);
        int axis = 0;
        float ext = cb.hi[0] - cb.lo[0];
        for (int a = 1; a < 3; a++) if (cb.hi[a] - cb.lo[a] > ext) { ext = cb.hi[a] - cb.lo[a]; axis = a; }
        int mid = -1;
        Box lbox, rbox;
        if (ext > 0.f) {
            Box bin_box[kBins]; int bin_cnt[kBins];
            for (int b = 0; b < kBins; b++) { bin_box[b].reset(); bin_cnt[b] = 0; }
            float scale = (float)kBins / ext;
            auto bin_of = [&](const Ref& r) { int b = (int)((r.c[axis] - cb.lo[axis]) * scale); return std::min(std::max(b, 0), kBins - 1); };
            for (int i = first; i < first + count; i++) { int b = bin_of(refs[i]); bin_box[b].grow(refs[i].box); bin_cnt[b]++; }
            float r_area[kBins]; int r_cnt[kBins];
            Box acc; acc.reset(); int n = 0;
            for (int b = kBins - 1; b > 0; b--) { acc.grow(bin_box[b]); n += bin_cnt[b]; r_area[b] = acc.half_area(); r_cnt[b] = n; }
            acc.reset(); n = 0;
            float best = std::numeric_limits<float>::infinity(); int best_b = -1;
            float inv_parent = 1.0f / std::max(box.half_area(), 1e-20f);
            for (int b = 0; b < kBins - 1; b++) {
                acc.grow(bin_box[b]); n += bin_cnt[b];
                if (n == 0 || r_cnt[b + 1] == 0) continue;
                float cost = kTraverseCost + (acc.half_area() * (float)n + r_area[b + 1] * (float)r_cnt[b + 1]) * inv_parent;
                if (cost < best) { best = cost; best_b = b; }
            }
            if (best_b >= 0 && (count > kMaxLeaf || best < (float)count)) {
                Ref* it = std::stable_partition(refs + first, refs + first + count, [&](const Ref& r) { return bin_of(r) <= best_b; });
                mid = (int)(it - refs);
            }
        }
        if (mid < 0) {
            if (count <= kMaxLeaf) return leaf_link(first, count);
            mid = first + count / 2;         // degenerate centroids: split by index
            std::stable_sort(refs + first, refs + first + count, [&](const Ref& a, const Ref& b) { return a.c[axis] < b.c[axis]; });
        }
        lbox.reset(); rbox.reset();
        for (int i = first; i < mid; i++) lbox.grow(refs[i].box);
        for (int i = mid; i < first + count; i++) rbox.grow(refs[i].box);
        int me = (int)(nodes.size() / 16);
        nodes.resize(nodes.size() + 16, 0.f);
        int32_t l = build(first, mid - first, lbox, depth + 1, me, 0);
        int32_t r = build(mid, first + count - mid, rbox, depth + 1, me, 1);
        float* nd = nodes.data() + 16 * (size_t)me;
        for (int a = 0; a < 3; a++) { nd[a] = lbox.lo[a]; nd[3 + a] = lbox.hi[a]; nd[6 + a] = rbox.lo[a]; nd[9 + a] = rbox.hi[a]; }
        nd[12] = as_float(l); nd[13] = as_float(r);
        return me;
    }
};
}  // namespace

int build_bvh(const float* prims, int n_prims, const int32_t* obj_info, int n_objects, BvhData& out, int max_leaf) {
    if (n_prims <= 0 || !prims || !obj_info || max_leaf < 1 || max_leaf > 15) return -1;
    std::vector<Ref> refs((size_t)n_prims);
    Builder b;
    b.kMaxLeaf = max_leaf;
    b.refs = refs.data();
    std::vector<uint8_t> is_sphere((size_t)n_prims, 0);
    for (int o = 0; o < n_objects; o++)
        for (int p = obj_info[3 * o]; p < obj_info[3 * o] + obj_info[3 * o + 1]; p++)
            if (p >= 0 && p < n_prims) is_sphere[(size_t)p] = obj_info[3 * o + 2] != 0;
    for (int p = 0; p < n_prims; p++) {
        const float* v = prims + 9 * (size_t)p;
        Ref& r = refs[(size_t)p];
        r.prim = p;
        r.box.reset();
        if (is_sphere[(size_t)p]) {
            for (int a = 0; a < 3; a++) { r.box.lo[a] = v[a] - v[3]; r.box.hi[a] = v[a] + v[3]; r.c[a] = v[a]; }
        } else {
            r.box.grow(v); r.box.grow(v + 3); r.box.grow(v + 6);
            for (int a = 0; a < 3; a++) r.c[a] = (v[a] + v[3 + a] + v[6 + a]) * (1.0f / 3.0f);
        }
        // conservative padding: computed hit points sit a few ulp off the primitive's plane
        for (int a = 0; a < 3; a++) {
            float pad = 1e-4f + 1e-5f * std::max(std::fabs(r.box.lo[a]), std::fabs(r.box.hi[a]));
            r.box.lo[a] -= pad; r.box.hi[a] += pad;
        }
    }
    Box root; root.reset();
    for (auto& r : refs) root.grow(r.box);
    // Large scenes: the top of the tree is built here, subtrees of at most n / 64 primitives by worker threads (same splits, same tree:
    // a subtree's build only reads and permutes its own primitive range), then the pieces are appended and their links rebased.
    const int threads = host_threads();
    std::vector<Task> tasks;
    if (threads > 1 && n_prims >= 65536) { b.tasks = &tasks; b.grain = std::max(4096, n_prims / 64); }
    int32_t link = b.build(0, n_prims, root, 0);
    if (!tasks.empty()) {
        std::vector<Builder> sub(tasks.size());
        std::vector<int32_t> sub_link(tasks.size(), 0);
        std::atomic<size_t> next_task{0};
        auto worker = [&]() {
            for (size_t t = next_task.fetch_add(1); t < tasks.size(); t = next_task.fetch_add(1)) {
                Builder& w = sub[t];
                w.refs = refs.data(); w.kMaxLeaf = max_leaf;
                sub_link[t] = w.build(tasks[t].first, tasks[t].count, tasks[t].box, tasks[t].depth);
            }
        };
        std::vector<std::thread> pool;
        const int n_workers = (int)std::min<size_t>((size_t)threads, tasks.size());
        for (int k = 0; k < n_workers; k++) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
        for (size_t t = 0; t < tasks.size(); t++) {
            const int32_t offset = (int32_t)(b.nodes.size() / 16);
            Builder& w = sub[t];
            for (size_t k = 0; k < w.nodes.size() / 16; k++)
                for (int c = 0; c < 2; c++) {
                    int32_t l; std::memcpy(&l, &w.nodes[16 * k + 12 + (size_t)c], 4);
                    if (l >= 0) { l += offset; std::memcpy(&w.nodes[16 * k + 12 + (size_t)c], &l, 4); }
                }
            b.nodes.insert(b.nodes.end(), w.nodes.begin(), w.nodes.end());
            const int32_t l = (sub_link[t] >= 0) ? sub_link[t] + offset : sub_link[t];
            b.nodes[16 * (size_t)tasks[t].parent + 12 + (size_t)tasks[t].slot] = Builder::as_float(l);
            b.max_depth = std::max(b.max_depth, w.max_depth);
            std::vector<float>().swap(w.nodes);
        }
    }
    if (link < 0) {
        // the whole scene is one leaf: wrap it so that node 0 always exists
        b.nodes.assign(16, 0.f);
        float* nd = b.nodes.data();
        Box empty; empty.reset();
        for (int a = 0; a < 3; a++) { nd[a] = root.lo[a]; nd[3 + a] = root.hi[a]; nd[6 + a] = empty.lo[a]; nd[9 + a] = empty.hi[a]; }
        nd[12] = Builder::as_float(link);
        nd[13] = Builder::as_float(~0);          // leaf with zero primitives
    }
    out.nodes = std::move(b.nodes);
    out.prim_order.resize((size_t)n_prims);
    for (int i = 0; i < n_prims; i++) out.prim_order[(size_t)i] = refs[(size_t)i].prim;
    out.max_depth = b.max_depth;
    return 0;
}

}  // namespace apt
