// bvh_build.hpp — host BVH builder interface (see bvh_build.cpp).
#pragma once
#include <cstdint>
#include <vector>

namespace apt {
struct BvhData {
    std::vector<float> nodes;          // 16 floats per node, node 0 = root
    std::vector<int32_t> prim_order;   // BVH-order slot -> original primitive index
    int max_depth = 0;
    int n_nodes() const { return (int)(nodes.size() / 16); }
};
// prims: n_prims*9 (triangle v0 v1 v2 | sphere centre, r r r, -); obj_info: n_objects*3 (first, count, is_sphere)
int build_bvh(const float* prims, int n_prims, const int32_t* obj_info, int n_objects, BvhData& out);
}  // namespace apt
