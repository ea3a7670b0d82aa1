// bvh_build.hpp — host BVH builder interface (see bvh_build.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

#ifndef APT_NODE_BYTES
#define APT_NODE_BYTES 64u
#endif
#define APT_NODE_DWORDS (APT_NODE_BYTES / 4u)
namespace apt {
// Host threads for the per-primitive and per-node loops of scene creation (tree build, collapse, record packing): the machine's, at most
// 32, or APT_HOST_THREADS.  Results never depend on the count.
inline int host_threads() {
    int t = (int)std::thread::hardware_concurrency();
    t = std::max(1, std::min(t, 32));
    if (const char* e = std::getenv("APT_HOST_THREADS")) t = std::max(1, std::atoi(e));
    return t;
}
// f(i) for i in [0, n) on up to `threads` host threads (contiguous chunks); small ranges stay on the caller
template <class F>
void parallel_for(int n, int threads, F f) {
    if (threads <= 1 || n < 2048) { for (int i = 0; i < n; i++) f(i); return; }
    const int t = std::min(threads, (n + 511) / 512);
    std::vector<std::thread> pool;
    pool.reserve((size_t)t);
    for (int k = 0; k < t; k++) {
        const int a = (int)((long long)n * k / t), b = (int)((long long)n * (k + 1) / t);
        pool.emplace_back([a, b, &f]() { for (int i = a; i < b; i++) f(i); });
    }
    for (auto& th : pool) th.join();
}

struct BvhData {
    std::vector<float> nodes;          // 16 floats per node, node 0 = root
    std::vector<int32_t> prim_order;   // BVH-order slot -> original primitive index
    int max_depth = 0;
    int n_nodes() const { return (int)(nodes.size() / 16); }
};
// prims: n_prims*9 (triangle v0 v1 v2 | sphere centre, r r r, -); obj_info: n_objects*3 (first, count, is_sphere)
// max_leaf: primitives per leaf (1..4; the 8-wide tree below wants 1)
int build_bvh(const float* prims, int n_prims, const int32_t* obj_info, int n_objects, BvhData& out, int max_leaf = 4);

// bvh_gpu.hip: the same binary tree (single-primitive leaves) built on the device.  algo 0: LBVH (Morton sort + Karras' radix tree +
// bottom-up fit); algo 1: PLOC (Morton sort + rounds of nearest-neighbour merging by union area: SAH-class quality)
int build_bvh_gpu(const float* prims, int n_prims, const int32_t* obj_info, int n_objects, int device, BvhData& out, int algo = 0);

// 8-wide tree with quantised child boxes, what the traversal kernels walk (bvh_wide.cpp; layout in traverse.hpp).
struct WideFrame { float gmin[3], gstep[3]; };      // the global grid the node corners live on: world = gmin + gstep * grid (power-of-two steps)
struct WideBvhData {
    WideFrame frame;
    std::vector<uint32_t> nodes;       // APT_NODE_DWORDS (16 = 64 bytes) per node, node 0 = root, breadth-first
    std::vector<int32_t> prim_order;   // leaf-order slot -> original primitive index
    int max_depth = 0;                 // levels of 8-wide nodes
    int n_nodes() const { return (int)(nodes.size() / APT_NODE_DWORDS); }
};
int build_wide_bvh(const BvhData& bvh2, WideBvhData& out);

// bvh_linear.cpp: the four arrays `bvh_cpp.bvh_build` returns (tracer/bvh/bvh.cpp:274-296), preorder with subtree-skip offsets
struct LinearBvhData {
    std::vector<float> bvh_minmax;     // n_prims * 6, tree order
    std::vector<float> node_minmax;    // n_nodes * 6
    std::vector<int32_t> bvh_info;     // n_prims * 2: object, original primitive
    std::vector<int32_t> node_info;    // n_nodes * 3: first primitive, count, subtree size
    int n_nodes() const { return (int)(node_info.size() / 3); }
    int n_prims() const { return (int)(bvh_info.size() / 2); }
};
int build_linear_bvh(const float* prims, int n_prims, const int32_t* obj_prim_cnt, const int32_t* obj_is_sphere, int n_objects,
                     const float world_min[3], const float world_max[3], LinearBvhData& out);

// flat_build.cpp: the records of the flat sweep (traverse.hpp FlatScene): precomputed-transform planar primitives (parallelograms merged),
// spheres, and the per-record table that maps a winning record back to its triangle and barycentrics
void planar_rows(const float* tri9, float out12[12]);      // one triangle: corner + rows U, V, T of [e1 e2 n]^-1 (the product build's BVH leaf record)
int build_flat(const float* prims, int n_prims, const int32_t* obj_info, int n_objects, const int32_t* prim_class, const uint8_t* transmissive,
               std::vector<float>& stream, std::vector<float>& tab, int counts[7]);      // counts: parallelograms, convex quads, triangles - each plain, then in coplanar groups - and spheres
}  // namespace apt
