// bvh_linear.cpp — `bvh_cpp.bvh_build`-compatible BVH builder (host).
//
// AdaPT's Taichi traversal (`PathTracer.ray_intersect_bvh`, tracer/path_tracer.py:338-422) walks a PREORDER-LINEARISED tree
// whose nodes carry a subtree-skip offset, and reads it from four flat arrays (tracer/path_tracer.py:155-170) that the native
// module `bvh_cpp.bvh_build(obj_array, obj_info, world_min, world_max)` returns (tracer/bvh/bvh.cpp:274-296):
//     float[N*6]  per-primitive box  (min xyz, max xyz), in tree order
//     float[M*6]  per-node box
//     int  [N*2]  (object index, original primitive index) per tree-order primitive
//     int  [M*3]  (first primitive, primitive count, subtree size) per node; leaf <=> subtree size == 1
// This file produces exactly those arrays so that an AdaPT checkout can import the MI355X library in place of its pybind11
// module (adapt_amd/bvh_cpp.py).  The renderer in this library does NOT walk this tree — its own traversal wants
// the two-boxes-per-node layout of bvh_build.cpp; this is the drop-in for the reference's own kernels.
//
// Behaviour follows the reference builder's decisions (tracer/bvh/bvh.cpp:19-212, bvh_helper.h:18-120):
//   * primitive box: triangle = min/max over its vertices, each axis thinner than 1e-4 widened by 1e-4 both ways;
//     sphere = centre -+ radius; centroid: triangle = vertex mean, sphere = centre
//   * split axis = largest extent of the CENTROID bounds; 12 bins over [min - 0.001, max + 0.001] of that axis
//   * more than 4 primitives: binned SAH, cost = 0.1 + (n_l A_l + n_r A_r) / A_node, accepted when cost < n; the primitives
//     are partitioned about the winning bin edge.  4 or fewer: median split (nth_element), accepted on the same cost test
//   * a rejected split makes a leaf of several primitives; single primitives are leaves; the root box is the world box
//   * nodes are emitted in preorder, subtree size = 1 + sizes of both children
// The reference is built with libstdc++ (tracer/setup.py: g++ -O3): `std::partition` and `std::nth_element` are used here as
// there, so primitive order inside a subtree — which the standard leaves unspecified — comes out the same.  Written directly in
// linear form: the recursion appends a node, recurses, and patches the subtree size (no intermediate pointer tree).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <vector>

#include "bvh_build.hpp"

namespace apt {
namespace {

constexpr int kSahBins = 12;
constexpr float kWalkCost = 0.1f;

struct Box3 {
    float lo[3], hi[3];
    Box3() { clear(); }
    void clear() { for (int a = 0; a < 3; a++) { lo[a] = 1e4f; hi[a] = -1e4f; } }     // the reference's "empty" box (bvh_helper.h:22-25)
    void merge(const Box3& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(b.lo[a], lo[a]); hi[a] = std::max(b.hi[a], hi[a]); } }
    float area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (float)(2. * (double)(dx * dy + dy * dz + dx * dz));                      // `2. *` is a double product upstream
    }
};

struct PrimRef { Box3 box; float c[3]; int32_t prim, obj; };

struct LinearBuilder {
    std::vector<PrimRef> refs;
    std::vector<float> node_box;       // 6 per node
    std::vector<int32_t> node_info;    // 3 per node

    int emit(int first, int count, const Box3& b) {
        const int me = (int)(node_info.size() / 3);
        for (int a = 0; a < 3; a++) node_box.push_back(b.lo[a]);
        for (int a = 0; a < 3; a++) node_box.push_back(b.hi[a]);
        node_info.push_back(first); node_info.push_back(count); node_info.push_back(1);
        return me;
    }

    // node over refs[first, first + count) with bounds `bound`; returns its subtree size
    int build(int first, int count, const Box3& bound) {
        const int me = emit(first, count, bound);
        if (count <= 1) return 1;
        const int last = first + count;
        // split axis: widest centroid extent
        float cmin[3], cmax[3];
        for (int a = 0; a < 3; a++) cmin[a] = cmax[a] = refs[(size_t)first].c[a];
        for (int i = first + 1; i < last; i++)
            for (int a = 0; a < 3; a++) { cmin[a] = std::min(cmin[a], refs[(size_t)i].c[a]); cmax[a] = std::max(cmax[a], refs[(size_t)i].c[a]); }
        int axis = 0;
        float extent = cmax[0] - cmin[0];
        for (int a = 1; a < 3; a++) if (cmax[a] - cmin[a] > extent) { extent = cmax[a] - cmin[a]; axis = a; }
        std::array<float, kSahBins> edge;                        // upper edges of the bins
        const float start = cmin[axis] - 0.001f, width = (extent + 0.002f) / (float)kSahBins;
        for (int b = 0; b < kSahBins; b++) edge[(size_t)b] = start + width * (float)(b + 1);

        const float n_total = (float)count;
        const float inv_area = (float)(1. / (double)bound.area());
        Box3 left, right;
        int n_left = 0;
        if (count > 4) {
            std::array<Box3, kSahBins> bin_box; std::array<int, kSahBins> bin_n{};
            for (int i = first; i < last; i++) {
                size_t b = (size_t)(std::lower_bound(edge.begin(), edge.end(), refs[(size_t)i].c[axis]) - edge.begin());
                if (b >= (size_t)kSahBins) b = kSahBins - 1;     // unreachable: the last edge lies 0.001 beyond the largest centroid
                bin_box[b].merge(refs[(size_t)i].box); bin_n[b]++;
            }
            std::array<float, kSahBins> area_l{}, area_r{}; std::array<int, kSahBins> cum{};
            Box3 fwd, bwd;
            int running = 0;
            for (int b = 0; b < kSahBins; b++) {
                fwd.merge(bin_box[(size_t)b]); running += bin_n[(size_t)b];
                cum[(size_t)b] = running; area_l[(size_t)b] = fwd.area();
                if (b > 0) { bwd.merge(bin_box[(size_t)(kSahBins - b)]); area_r[(size_t)(kSahBins - 1 - b)] = bwd.area(); }
            }
            float best = 5e9f; int best_b = 0;
            for (int b = 0; b < kSahBins - 1; b++) {
                const float cost = kWalkCost + inv_area * ((float)cum[(size_t)b] * area_l[(size_t)b] + (n_total - (float)cum[(size_t)b]) * area_r[(size_t)b]);
                if (cost < best) { best = cost; best_b = b; }
            }
            if (best < n_total) {
                const float pivot = edge[(size_t)best_b];
                std::partition(refs.begin() + first, refs.begin() + last, [pivot, axis](const PrimRef& r) { return r.c[axis] < pivot; });
                n_left = cum[(size_t)best_b];
            }
            for (int b = 0; b <= best_b; b++) left.merge(bin_box[(size_t)b]);
            for (int b = kSahBins - 1; b > best_b; b--) right.merge(bin_box[(size_t)b]);
        } else {
            const int mid = (first + last) >> 1;
            std::nth_element(refs.begin() + first, refs.begin() + mid, refs.begin() + last,
                             [axis](const PrimRef& a, const PrimRef& b) { return a.c[axis] < b.c[axis]; });
            for (int i = first; i < mid; i++) left.merge(refs[(size_t)i].box);
            for (int i = mid; i < last; i++) right.merge(refs[(size_t)i].box);
            n_left = mid - first;
            const float cost = kWalkCost + inv_area * (left.area() * (float)n_left + right.area() * (n_total - (float)n_left));
            if (cost >= n_total) n_left = 0;
        }
        if (n_left <= 0) return 1;                               // several primitives, no worthwhile split: a fat leaf
        int size = 1;
        size += build(first, n_left, left);
        size += build(first + n_left, count - n_left, right);
        node_info[3 * (size_t)me + 2] = size;
        return size;
    }
};

}  // namespace

int build_linear_bvh(const float* prims, int n_prims, const int32_t* obj_prim_cnt, const int32_t* obj_is_sphere, int n_objects,
                     const float world_min[3], const float world_max[3], LinearBvhData& out) {
    if (!prims || !obj_prim_cnt || !obj_is_sphere || !world_min || !world_max || n_prims <= 0 || n_objects <= 0) return -1;
    long total = 0;
    for (int o = 0; o < n_objects; o++) { if (obj_prim_cnt[o] < 0) return -1; total += obj_prim_cnt[o]; }
    if (total != n_prims) return -1;
    LinearBuilder b;
    b.refs.resize((size_t)n_prims);
    int p = 0;
    for (int o = 0; o < n_objects; o++)
        for (int k = 0; k < obj_prim_cnt[o]; k++, p++) {
            const float* v = prims + 9 * (size_t)p;
            PrimRef& r = b.refs[(size_t)p];
            r.prim = p; r.obj = o;
            if (obj_is_sphere[o] > 0) {
                for (int a = 0; a < 3; a++) { r.box.lo[a] = v[a] - v[3 + a]; r.box.hi[a] = v[a] + v[3 + a]; r.c[a] = v[a]; }
            } else {
                for (int a = 0; a < 3; a++) {
                    r.box.lo[a] = std::min(std::min(v[a], v[3 + a]), v[6 + a]);
                    r.box.hi[a] = std::max(std::max(v[a], v[3 + a]), v[6 + a]);
                    // bvh_helper.h:38-42: the float difference is compared with the DOUBLE literal 1e-4 and the bounds move in double before they are stored as float
                    if ((double)(r.box.hi[a] - r.box.lo[a]) < 1e-4) { r.box.lo[a] = (float)((double)r.box.lo[a] - 1e-4); r.box.hi[a] = (float)((double)r.box.hi[a] + 1e-4); }
                    // vertex mean as Eigen 3.4's fixed-size reduction evaluates it for three terms: x0 + (x1 + x2), then / 3
                    r.c[a] = (v[a] + (v[3 + a] + v[6 + a])) / 3.0f;
                }
            }
        }
    Box3 root;
    for (int a = 0; a < 3; a++) { root.lo[a] = world_min[a]; root.hi[a] = world_max[a]; }
    b.node_box.reserve((size_t)n_prims * 12); b.node_info.reserve((size_t)n_prims * 6);
    b.build(0, n_prims, root);
    out.node_minmax = std::move(b.node_box);
    out.node_info = std::move(b.node_info);
    out.bvh_minmax.resize((size_t)n_prims * 6); out.bvh_info.resize((size_t)n_prims * 2);
    for (int i = 0; i < n_prims; i++) {
        const PrimRef& r = b.refs[(size_t)i];
        for (int a = 0; a < 3; a++) { out.bvh_minmax[6 * (size_t)i + a] = r.box.lo[a]; out.bvh_minmax[6 * (size_t)i + 3 + a] = r.box.hi[a]; }
        out.bvh_info[2 * (size_t)i] = r.obj; out.bvh_info[2 * (size_t)i + 1] = r.prim;
    }
    return 0;
}

}  // namespace apt
