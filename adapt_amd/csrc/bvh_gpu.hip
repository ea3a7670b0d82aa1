// bvh_gpu.hip — binary BVH built on the GPU (PLOC or LBVH), the alternative to the host SAH builder of bvh_build.cpp.
//
// Replaces the same reference component, the recursive single-threaded SAH build of `bvh_cpp.bvh_build`
// (tracer/bvh/bvh.cpp:83-212), where scene-load time matters more than the last per cent of tree quality ("removes
// the CPU build from scene-load time for 290k+ prims").  All O(n log n) work runs on the device:
//   1. k_prim_boxes    padded primitive boxes (the host builder's padding) + centroids, centroid bounds by ordered-int atomics
//   2. k_morton        30-bit Morton code of the centroid, made unique by the primitive index in the low word (64-bit key)
//   3. hipcub radix sort of the keys (only the 30 + ceil(log2 n) significant bits)
// then either (algo 1, the default) PLOC:
//   4. k_ploc_*        rounds of: nearest neighbour by union area within +-16 places, merge of mutual pairs, compaction (see below)
//   5. k_ploc_emit     the two-boxes-per-node record of bvh_build.cpp, single-primitive leaves, root at node 0
// or (algo 0) LBVH:
//   4. k_radix_tree    Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees": every inner
//                      node finds its key range and split independently from the common-prefix lengths of neighbouring keys
//   5. k_fit           leaves walk up, the second child to arrive at a node merges the two child boxes (one atomic counter per node)
//   6. k_emit          the exported record
// The result is downloaded in the layout of apt::BvhData, so that everything downstream (collapse to the 8-wide quantised tree,
// primitive records in leaf order) is shared with the SAH path.  Closest-hit results do not depend on the tree (exact
// per-primitive tests, exact tie-break); only the amount of work per ray does.  Measured (api.hip, builder choice): rendering on the
// PLOC tree is 1-5 % slower than on the binned-SAH tree, on the radix tree 9-14 %; SAH stays the default below a million primitives.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "bvh_build.hpp"

namespace apt {
namespace {

#define GB 256

__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return (i >= 0) ? i : (i ^ 0x7fffffff); }     // monotone float -> int
__device__ __forceinline__ float ord2f(int i) { return __int_as_float((i >= 0) ? i : (i ^ 0x7fffffff)); }

__global__ void k_prim_boxes(const float* __restrict__ prims, const uint8_t* __restrict__ is_sphere, int n, float* __restrict__ boxes, float* __restrict__ cent, int* bounds) {
    const int p = blockIdx.x * GB + threadIdx.x;
    float c[3] = {0.f, 0.f, 0.f};
    if (p < n) {
        const float* v = prims + 9 * (size_t)p;
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            if (is_sphere[p]) { lo[a] = v[a] - v[3]; hi[a] = v[a] + v[3]; c[a] = v[a]; }
            else { lo[a] = fminf(fminf(v[a], v[3 + a]), v[6 + a]); hi[a] = fmaxf(fmaxf(v[a], v[3 + a]), v[6 + a]); c[a] = (v[a] + v[3 + a] + v[6 + a]) * (1.0f / 3.0f); }
            const float pad = 1e-4f + 1e-5f * fmaxf(fabsf(lo[a]), fabsf(hi[a]));      // bvh_build.cpp: hit points sit a few ulp off the primitive's plane
            boxes[6 * (size_t)p + a] = lo[a] - pad; boxes[6 * (size_t)p + 3 + a] = hi[a] + pad;
            cent[3 * (size_t)p + a] = c[a];
        }
    }
    // centroid bounds: wave reduction, then one ordered-int atomic per wave and axis
    for (int a = 0; a < 3; a++) {
        float mn = (p < n) ? c[a] : 3.0e38f, mx = (p < n) ? c[a] : -3.0e38f;
        for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_down(mn, off)); mx = fmaxf(mx, __shfl_down(mx, off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&bounds[a], f2ord(mn)); atomicMax(&bounds[3 + a], f2ord(mx)); }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {       // 10 bits -> every third bit
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000ffu; x = (x | (x << 8)) & 0x0300f00fu; x = (x | (x << 4)) & 0x030c30c3u; x = (x | (x << 2)) & 0x09249249u;
    return x;
}
__global__ void k_morton(const float* __restrict__ cent, int n, const int* __restrict__ bounds, unsigned long long* __restrict__ keys) {
    const int p = blockIdx.x * GB + threadIdx.x;
    if (p >= n) return;
    uint32_t code = 0;
    for (int a = 0; a < 3; a++) {
        const float lo = ord2f(bounds[a]), hi = ord2f(bounds[3 + a]);
        const float ext = hi - lo;
        float q = (ext > 0.f) ? (cent[3 * (size_t)p + a] - lo) / ext : 0.f;
        q = fminf(fmaxf(q * 1024.f, 0.f), 1023.f);
        code |= spread10((uint32_t)q) << (2 - a);
    }
    keys[p] = ((unsigned long long)code << 32) | (unsigned long long)(uint32_t)p;
}

__device__ __forceinline__ int prefix(const unsigned long long* __restrict__ keys, int n, int i, int j) {      // common-prefix length, -1 outside
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));            // keys are unique (index in the low word): never 64
}
// inner node i of n - 1: children as links (inner index, or ~leaf position for a leaf), parents of both children
__global__ void k_radix_tree(const unsigned long long* __restrict__ keys, int n, int* __restrict__ left, int* __restrict__ right, int* __restrict__ parent_inner, int* __restrict__ parent_leaf) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = prefix(keys, n, i, i - d);
    int lmax = 2;
    while (prefix(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = prefix(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int g = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int l_link = (lo == g) ? ~g : g, r_link = (hi == g + 1) ? ~(g + 1) : g + 1;
    left[i] = l_link; right[i] = r_link;
    if (l_link < 0) parent_leaf[g] = i; else parent_inner[g] = i;
    if (r_link < 0) parent_leaf[g + 1] = i; else parent_inner[g + 1] = i;
}

// boxes of the inner nodes, bottom up: the second child to arrive at a node computes it and moves on
__global__ void k_fit(int n, const unsigned long long* __restrict__ keys, const float* __restrict__ prim_box, const int* __restrict__ left, const int* __restrict__ right,
                      const int* __restrict__ parent_inner, const int* __restrict__ parent_leaf, float* node_box, int* arrived) {
    const int s = blockIdx.x * GB + threadIdx.x;
    if (s >= n) return;
    int node = parent_leaf[s];
    while (node >= 0) {
        if (atomicAdd(&arrived[node], 1) == 0) return;          // first child: the sibling will do the work
        __threadfence();
        float lo[3], hi[3];
        for (int c = 0; c < 2; c++) {
            const int link = c ? right[node] : left[node];
            const float* b = (link < 0) ? prim_box + 6 * (size_t)(uint32_t)(keys[~link] & 0xffffffffull) : node_box + 6 * (size_t)link;
            for (int a = 0; a < 3; a++) {
                const float bl = (link < 0) ? b[a] : __builtin_nontemporal_load(&b[a]), bh = (link < 0) ? b[3 + a] : __builtin_nontemporal_load(&b[3 + a]);
                lo[a] = c ? fminf(lo[a], bl) : bl; hi[a] = c ? fmaxf(hi[a], bh) : bh;
            }
        }
        for (int a = 0; a < 3; a++) { node_box[6 * (size_t)node + a] = lo[a]; node_box[6 * (size_t)node + 3 + a] = hi[a]; }
        __threadfence();
        node = (node == 0) ? -1 : parent_inner[node];
    }
}

// the exported record of bvh_build.cpp: [0..5] left box, [6..11] right box, [12] left link, [13] right link; leaf link = ~(slot << 4 | 1)
__global__ void k_emit(int n, const unsigned long long* __restrict__ keys, const float* __restrict__ prim_box, const float* __restrict__ node_box,
                       const int* __restrict__ left, const int* __restrict__ right, float* __restrict__ out, int* __restrict__ prim_order) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i < n) prim_order[i] = (int)(uint32_t)(keys[i] & 0xffffffffull);
    if (i >= n - 1) return;
    float* nd = out + 16 * (size_t)i;
    for (int c = 0; c < 2; c++) {
        const int link = c ? right[i] : left[i];
        const float* b = (link < 0) ? prim_box + 6 * (size_t)(uint32_t)(keys[~link] & 0xffffffffull) : node_box + 6 * (size_t)link;
        for (int a = 0; a < 6; a++) nd[6 * c + a] = b[a];
        nd[12 + c] = __int_as_float((link < 0) ? ~(((~link) << 4) | 1) : link);
    }
    nd[14] = 0.f; nd[15] = 0.f;
}

// ---------------------------------------------------------------- PLOC (parallel locally-ordered clustering)
// Meister, Bittner, "Parallel locally-ordered clustering for bounding volume hierarchy construction" (TVCG 2018): bottom-up
// agglomeration over the Morton-ordered cluster array.  Every round each cluster looks `radius` places to either side for the
// neighbour whose union with it has the smallest surface area; clusters that chose each other merge into a new node; the array is
// compacted (prefix sum) and the next round starts, until one cluster is left.  Unlike the radix tree above, where a split is
// decided by key bits alone, every merge here is decided by box areas - the quantity the SAH prices - so the tree walks like
// a SAH tree while the whole build still runs on the device.
// Ties go to the aligned partner i ^ 1, then to the lower position, which guarantees a mutual pair in every round (k_ploc_nn), so the
// loop always terminates; should a build fail all the same, the caller falls back to the radix tree and then to the host builder.
#define PLOC_RADIUS 16
__device__ __forceinline__ float union_half_area(const float* __restrict__ a, const float* __restrict__ b) {
    const float d0 = fmaxf(a[3], b[3]) - fminf(a[0], b[0]), d1 = fmaxf(a[4], b[4]) - fminf(a[1], b[1]), d2 = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return d0 * d1 + d1 * d2 + d0 * d2;
}
__global__ void k_ploc_init(int n, const unsigned long long* __restrict__ keys, const float* __restrict__ prim_box, float* __restrict__ cbox, int* __restrict__ clink) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i >= n) return;
    const float* b = prim_box + 6 * (size_t)(uint32_t)(keys[i] & 0xffffffffull);
    for (int a = 0; a < 6; a++) cbox[6 * (size_t)i + a] = b[a];
    clink[i] = ~i;                                                  // leaf = ~(position in Morton order)
}
__global__ void k_ploc_nn(int m, const float* __restrict__ cbox, int* __restrict__ nn) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i >= m) return;
    float mine[6];
    for (int a = 0; a < 6; a++) mine[a] = cbox[6 * (size_t)i + a];
    float best = 3.0e38f; int best_j = -1;
    const int lo = max(0, i - PLOC_RADIUS), hi = min(m - 1, i + PLOC_RADIUS);
    for (int j = lo; j <= hi; j++) {                                // ascending: of equal areas the lower position stays ...
        if (j == i) continue;
        const float d = union_half_area(mine, cbox + 6 * (size_t)j);
        if (d < best) { best = d; best_j = j; }
    }
    // ... unless the aligned partner i ^ 1 ties with the best: duplicated primitives or a ribbon of equal triangles make EVERY area equal,
    // and "the lower neighbour" then forms a chain i -> i - 1 -> i - 2 ... with one mutual pair per round (rounds linear in n: 4 095 for
    // 4 096 identical triangles).  Preferring i ^ 1 pairs such runs off two by two; a mutual pair still exists in every round (an aligned
    // tie is mutual by itself, and without one the argument for the lowest pair holds as before).
    const int pj = i ^ 1;
    if (pj >= lo && pj <= hi && pj < m && union_half_area(mine, cbox + 6 * (size_t)pj) == best) best_j = pj;
    nn[i] = best_j;
}
// flags[i]: low word 1 = cluster i survives the round (alone, or as the merged cluster), high word 1 = it is the lower half of a merging pair
__global__ void k_ploc_mark(int m, const int* __restrict__ nn, unsigned long long* __restrict__ flags) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i >= m) return;
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == i;
    flags[i] = (mutual && i > j) ? 0ull : (1ull | ((mutual ? 1ull : 0ull) << 32));
}
__global__ void k_ploc_apply(int m, int node_base, const int* __restrict__ nn, const unsigned long long* __restrict__ flags, const unsigned long long* __restrict__ pos,
                             const float* __restrict__ cbox, const int* __restrict__ clink, float* __restrict__ cbox2, int* __restrict__ clink2,
                             int* __restrict__ left, int* __restrict__ right, float* __restrict__ node_box) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i >= m) return;
    const unsigned long long f = flags[i];
    if (!(f & 1ull)) return;
    const int p = (int)(uint32_t)(pos[i] & 0xffffffffull);
    float box[6];
    for (int a = 0; a < 6; a++) box[a] = cbox[6 * (size_t)i + a];
    int link = clink[i];
    if (f >> 32) {
        const int j = nn[i], node = node_base + (int)(uint32_t)(pos[i] >> 32);
        const float* o = cbox + 6 * (size_t)j;
        for (int a = 0; a < 3; a++) { box[a] = fminf(box[a], o[a]); box[3 + a] = fmaxf(box[3 + a], o[3 + a]); }
        left[node] = link; right[node] = clink[j];
        for (int a = 0; a < 6; a++) node_box[6 * (size_t)node + a] = box[a];
        link = node;
    }
    for (int a = 0; a < 6; a++) cbox2[6 * (size_t)p + a] = box[a];
    clink2[p] = link;
}
// nodes were numbered in creation order, the root last: the exported tree wants the root at 0 -> index n - 2 - k
__global__ void k_ploc_emit(int n, const unsigned long long* __restrict__ keys, const float* __restrict__ prim_box, const float* __restrict__ node_box,
                            const int* __restrict__ left, const int* __restrict__ right, float* __restrict__ out, int* __restrict__ prim_order) {
    const int i = blockIdx.x * GB + threadIdx.x;
    if (i < n) prim_order[i] = (int)(uint32_t)(keys[i] & 0xffffffffull);
    if (i >= n - 1) return;
    float* nd = out + 16 * (size_t)(n - 2 - i);
    for (int c = 0; c < 2; c++) {
        const int link = c ? right[i] : left[i];
        const float* b = (link < 0) ? prim_box + 6 * (size_t)(uint32_t)(keys[~link] & 0xffffffffull) : node_box + 6 * (size_t)link;
        for (int a = 0; a < 6; a++) nd[6 * c + a] = b[a];
        nd[12 + c] = __int_as_float((link < 0) ? ~(((~link) << 4) | 1) : (n - 2 - link));
    }
    nd[14] = 0.f; nd[15] = 0.f;
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
    template <class T> T* as() const { return (T*)p; }
};
#define GTRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return -10 - (int)e_; } while (0)

}  // namespace

static int build_bvh_gpu_impl(const float* prims, int n, const int32_t* obj_info, int n_objects, int device, BvhData& out, int algo) {
    if (!prims || !obj_info || n < 2) return -1;
    if ((long long)n >= (1ll << 27)) return -1;                      // leaf links keep the slot in 27 bits
    GTRY(hipSetDevice(device));
    std::vector<uint8_t> sph((size_t)n, 0);
    for (int o = 0; o < n_objects; o++)
        for (int p = obj_info[3 * o]; p < obj_info[3 * o] + obj_info[3 * o + 1]; p++)
            if (p >= 0 && p < n) sph[(size_t)p] = obj_info[3 * o + 2] != 0;
    Buf d_prims, d_sph, d_box, d_cent, d_bounds, d_keys, d_keys2, d_tmp, d_left, d_right, d_pi, d_pl, d_nbox, d_arr, d_out, d_order;
    GTRY(d_prims.alloc((size_t)n * 36)); GTRY(d_sph.alloc((size_t)n)); GTRY(d_box.alloc((size_t)n * 24)); GTRY(d_cent.alloc((size_t)n * 12));
    GTRY(d_bounds.alloc(24)); GTRY(d_keys.alloc((size_t)n * 8)); GTRY(d_keys2.alloc((size_t)n * 8));
    GTRY(d_left.alloc((size_t)n * 4)); GTRY(d_right.alloc((size_t)n * 4)); GTRY(d_pi.alloc((size_t)n * 4)); GTRY(d_pl.alloc((size_t)n * 4));
    GTRY(d_nbox.alloc((size_t)n * 24)); GTRY(d_arr.alloc((size_t)n * 4)); GTRY(d_out.alloc((size_t)(n - 1) * 64)); GTRY(d_order.alloc((size_t)n * 4));
    GTRY(hipMemcpy(d_prims.p, prims, (size_t)n * 36, hipMemcpyHostToDevice));
    GTRY(hipMemcpy(d_sph.p, sph.data(), (size_t)n, hipMemcpyHostToDevice));
    const int init_ord[6] = {0x7f7fffff, 0x7f7fffff, 0x7f7fffff, (int)0x80800000, (int)0x80800000, (int)0x80800000};      // f2ord(+FLT_MAX) x 3, f2ord(-FLT_MAX) x 3
    GTRY(hipMemcpy(d_bounds.p, init_ord, 24, hipMemcpyHostToDevice));
    GTRY(hipMemset(d_arr.p, 0, (size_t)n * 4));
    GTRY(hipMemset(d_pi.p, 0xff, (size_t)n * 4));
    const int grid = (n + GB - 1) / GB;
    hipLaunchKernelGGL(k_prim_boxes, dim3(grid), dim3(GB), 0, 0, d_prims.as<float>(), d_sph.as<uint8_t>(), n, d_box.as<float>(), d_cent.as<float>(), d_bounds.as<int>());
    hipLaunchKernelGGL(k_morton, dim3(grid), dim3(GB), 0, 0, d_cent.as<float>(), n, d_bounds.as<int>(), d_keys.as<unsigned long long>());
    // keys are born in index order and the radix sort is stable: sorting on the 30 Morton bits (32..61) alone orders them by (code, index)
    size_t tmp_bytes = 0;
    GTRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), n, 32, 62));
    GTRY(d_tmp.alloc(tmp_bytes));
    GTRY(hipcub::DeviceRadixSort::SortKeys(d_tmp.p, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), n, 32, 62));
    std::swap(d_keys.p, d_keys2.p);
    if (algo == 1) {
        // PLOC: d_left / d_right / d_nbox are indexed by node creation order here; d_pi, d_pl, d_arr are free to serve as round buffers
        Buf d_cb[2], d_cl[2], d_flags, d_pos, d_scan;
        for (int k = 0; k < 2; k++) { GTRY(d_cb[k].alloc((size_t)n * 24)); GTRY(d_cl[k].alloc((size_t)n * 4)); }
        GTRY(d_flags.alloc((size_t)n * 8)); GTRY(d_pos.alloc((size_t)n * 8));
        size_t scan_bytes = 0;
        GTRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_flags.as<unsigned long long>(), d_pos.as<unsigned long long>(), n));
        GTRY(d_scan.alloc(scan_bytes));
        hipLaunchKernelGGL(k_ploc_init, dim3(grid), dim3(GB), 0, 0, n, d_keys.as<unsigned long long>(), d_box.as<float>(), d_cb[0].as<float>(), d_cl[0].as<int>());
        int m = n, node_base = 0, cur = 0;
        for (int round = 0; m > 1; round++) {
            if (round > 4096) return -3;                                 // (every round merges at least one pair; the caller falls back to the radix tree)
            const int g = (m + GB - 1) / GB;
            hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(GB), 0, 0, m, d_cb[cur].as<float>(), d_pi.as<int>());
            hipLaunchKernelGGL(k_ploc_mark, dim3(g), dim3(GB), 0, 0, m, d_pi.as<int>(), d_flags.as<unsigned long long>());
            GTRY(hipcub::DeviceScan::ExclusiveSum(d_scan.p, scan_bytes, d_flags.as<unsigned long long>(), d_pos.as<unsigned long long>(), m));
            hipLaunchKernelGGL(k_ploc_apply, dim3(g), dim3(GB), 0, 0, m, node_base, d_pi.as<int>(), d_flags.as<unsigned long long>(), d_pos.as<unsigned long long>(),
                               d_cb[cur].as<float>(), d_cl[cur].as<int>(), d_cb[cur ^ 1].as<float>(), d_cl[cur ^ 1].as<int>(), d_left.as<int>(), d_right.as<int>(), d_nbox.as<float>());
            unsigned long long last_pos = 0, last_flag = 0;
            GTRY(hipMemcpy(&last_pos, d_pos.as<unsigned long long>() + (m - 1), 8, hipMemcpyDeviceToHost));
            GTRY(hipMemcpy(&last_flag, d_flags.as<unsigned long long>() + (m - 1), 8, hipMemcpyDeviceToHost));
            const unsigned long long tot = last_pos + last_flag;
            const int survivors = (int)(uint32_t)(tot & 0xffffffffull), merges = (int)(uint32_t)(tot >> 32);
            if (merges < 1 || survivors != m - merges) return -4;
            node_base += merges; m = survivors; cur ^= 1;
        }
        if (node_base != n - 1) return -5;
        hipLaunchKernelGGL(k_ploc_emit, dim3(grid), dim3(GB), 0, 0, n, d_keys.as<unsigned long long>(), d_box.as<float>(), d_nbox.as<float>(), d_left.as<int>(), d_right.as<int>(),
                           d_out.as<float>(), d_order.as<int>());
    } else {
        hipLaunchKernelGGL(k_radix_tree, dim3(grid), dim3(GB), 0, 0, d_keys.as<unsigned long long>(), n, d_left.as<int>(), d_right.as<int>(), d_pi.as<int>(), d_pl.as<int>());
        hipLaunchKernelGGL(k_fit, dim3(grid), dim3(GB), 0, 0, n, d_keys.as<unsigned long long>(), d_box.as<float>(), d_left.as<int>(), d_right.as<int>(), d_pi.as<int>(), d_pl.as<int>(),
                           d_nbox.as<float>(), d_arr.as<int>());
        hipLaunchKernelGGL(k_emit, dim3(grid), dim3(GB), 0, 0, n, d_keys.as<unsigned long long>(), d_box.as<float>(), d_nbox.as<float>(), d_left.as<int>(), d_right.as<int>(),
                           d_out.as<float>(), d_order.as<int>());
    }
    GTRY(hipGetLastError());
    GTRY(hipDeviceSynchronize());
    out.nodes.resize((size_t)(n - 1) * 16);
    out.prim_order.resize((size_t)n);
    GTRY(hipMemcpy(out.nodes.data(), d_out.p, (size_t)(n - 1) * 64, hipMemcpyDeviceToHost));
    GTRY(hipMemcpy(out.prim_order.data(), d_order.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    out.max_depth = 64;                 // a radix tree over 62-bit keys is at most that deep; nobody sizes anything by it (the 8-wide collapse recounts)
    return 0;
}
int build_bvh_gpu(const float* prims, int n, const int32_t* obj_info, int n_objects, int device, BvhData& out, int algo) {
    int rc = build_bvh_gpu_impl(prims, n, obj_info, n_objects, device, out, algo);
    if (rc != 0 && rc > -10 && algo == 1) rc = build_bvh_gpu_impl(prims, n, obj_info, n_objects, device, out, 0);      // PLOC gave up (-3 .. -5): the radix tree always builds
    return rc;
}

}  // namespace apt
