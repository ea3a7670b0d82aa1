// api.hip — host driver + C-ABI (include/adapt_mi.h) of the gfx950 wavefront path tracer.
//
// Everything the reference does between `Renderer.__init__` and `pixels.to_numpy()` on the
// Taichi side (field allocation + upload: tracer_base.py:76-134, path_tracer.py:245-274; one
// kernel launch per spp: render.py:118-122; readback: utils/watermark.py:23) happens here:
// scene upload, BVH build, queue allocation in HBM, the per-batch stage schedule on a private
// HIP stream, statistics and HIP-event timing.  Stage kernels live in stages.hpp, shade_stage.hpp and volumetric.hpp.
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/adapt_mi.h"
#include "bvh_build.hpp"
#include "shade_stage.hpp"
#include "unit_kernels.hpp"
#include "volumetric.hpp"

#define APT_EXPORT extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY_VOID(expr) do { hipError_t e__ = (expr); (void)e__; } while (0)
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(APT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

// ------------------------------------------------- shade kernel specialisations
// (material mask, emitter mask) -> instantiation; the host picks the first one that covers the scene
typedef void (*shade_fn)(DevScene, Params, Queues, Counters*, ShadeIn, int, int);
#if APT_FAST
#define APT_FUSED_FN(...) __VA_ARGS__
#else
#define APT_FUSED_FN(...) nullptr         // rays traced by the shade kernel: a product-build path (it rides on the flat sweep's records)
#endif
typedef void (*shade_traced_fn)(DevScene, Params, Queues, Counters*, int, int);
struct ShadeVariant { int bm, sm; shade_fn fn; const char* name; shade_traced_fn traced; };      // traced: the kernel that traces its light sample and its continuation ray itself (shade_stage.hpp k_shade_traced: flat sweep, one light sample per vertex)
static const ShadeVariant kShadeVariants[] = {
    {0x002, 0x01, k_shade<0x002, 0x01>, "lambertian/point", APT_FUSED_FN(k_shade_traced_lean<0x002, 0x01>)},
    {0x003, 0x03, k_shade<0x003, 0x03>, "phong+lambertian/point+area", APT_FUSED_FN(k_shade_traced<0x003, 0x03>)},
    {0x107, 0x03, k_shade<0x107, 0x03>, "phong+lambertian+mirror+glass/point+area", APT_FUSED_FN(k_shade_traced<0x107, 0x03>)},
    {APT_BX_ALL, APT_SRC_ALL, k_shade<APT_BX_ALL, APT_SRC_ALL>, "all models", APT_FUSED_FN(k_shade_traced<APT_BX_ALL, APT_SRC_ALL>)},
};
static const ShadeVariant kTexturedShade = {APT_BX_ALL, APT_SRC_ALL, k_shade<APT_BX_ALL, APT_SRC_ALL, 1>, "all models + image textures", APT_FUSED_FN(k_shade_traced<APT_BX_ALL, APT_SRC_ALL, 1>)};
// Material classes for sorted shading: (class mask) x (emitter mask: point+area | all)
// A kernel's register allocation is the maximum over the models it contains, so the classes are as fine as the models'
// footprints differ: Lambertian alone runs at 4 waves per SIMD, together with Blinn-Phong (three double pows) at 2-3.
#define APT_N_CLASS_DEFS 10
static const int kClassMask[APT_N_CLASS_DEFS] = {
    0x002,      // Lambertian
    0x001,      // Blinn-Phong
    0x040,      // Oren-Nayar
    0x504,      // delta: mirror BRDF, det-refraction BSDF, null BSDF
    0x010,      // modified Phong
    0x020,      // Fresnel blend
    0x080,      // thin coat
    0x200,      // Lambertian transmission
    0x008,      // Trowbridge-Reitz microfacet (upstream's opt-in model: type 3 reaches the device only with the switch on)
    0x801,      // Blinn-Phong objects without a specular lobe (k_s = 0, finite k_g >= 0: shading.hpp mask bit 11): no double-precision pow
};
static const char* kClassName[APT_N_CLASS_DEFS] = {"lambertian", "blinn-phong", "oren-nayar", "delta", "mod-phong", "fresnel-blend", "thin-coat", "lambert-trans", "microfacet", "blinn-phong(no lobe)"};
// Class kernels in groups (shade_stage.hpp k_shade_group): one launch per GROUP and bounce instead of one per class.  Groups follow the register
// footprints - a kernel allocates for its largest member: 0 = the lean classes, up to 96 VGPRs (five waves per SIMD), 1 = the rest, up to 128 (four).
typedef void (*group_fn)(DevScene, Params, Queues, Counters*, GroupIn, int, int);
#define APT_N_GROUPS 2
static const int kClassGroup[APT_N_CLASS_DEFS] = {0, 1, 1, 0, 1, 1, 1, 0, 1, 0};      // class definition -> group
static const int kClassSlot[APT_N_CLASS_DEFS] = {0, 0, 1, 1, 4, 5, 2, 2, 3, 3};       // ... and its member slot there (the B0..B5 order below)
#ifndef APT_GROUP0_WAVES
#define APT_GROUP0_WAVES 5
#endif
#define APT_GROUP_ROW(SM) {k_shade_group<SM, APT_GROUP0_WAVES, 0x002, 0x504, 0x200, 0x801>, k_shade_group<SM, 4, 0x001, 0x040, 0x080, 0x008, 0x010, 0x020>}
// (group 0 at five waves: 93 VGPRs with point + spot lights, 96 and one 8-byte scratch slot with area lights; at four it took 99)
static const group_fn kGroupShade[3][APT_N_GROUPS] = {APT_GROUP_ROW(0x03), APT_GROUP_ROW(APT_SRC_ALL), APT_GROUP_ROW(0x05)};      // [emitter set: point + area | all | point + spot (no area light: no emission code, no pdf in the record, both Philox blocks up front)][group]
#define APT_CLASS_PHONG 1
#define APT_CLASS_PHONG_NO_LOBE 9
static int class_of(int is_bsdf, int type, bool no_lobe) {
    const int bit = is_bsdf ? (type == 0 ? 8 : (type == 1 ? 9 : 10)) : (type & 7);
    if (!is_bsdf && (type & 7) == 0 && no_lobe) return APT_CLASS_PHONG_NO_LOBE;
    for (int c = 0; c < APT_N_CLASS_DEFS - 1; c++) if ((kClassMask[c] >> bit) & 1) return c;
    return 0;
}
typedef void (*extend_fn)(DevScene, Params, Queues, Counters*, int, const uint32_t*, LdsPlan);
typedef void (*shadow_fn)(DevScene, Params, Queues, Counters*, LdsPlan);
typedef void (*occluded_fn)(DevScene, uint32_t, const float*, const float*, const float*, int*, LdsPlan);
#if APT_FAST
#define APT_FLAT_FN(...) __VA_ARGS__
#else
#define APT_FLAT_FN(...) nullptr          // the flat sweep exists in the fast build only (traverse.hpp)
#endif
static const extend_fn kExtend[4][2] = {{k_extend<0, 0>, k_extend<0, 1>}, {k_extend<1, 0>, k_extend<1, 1>}, {k_extend<2, 0>, k_extend<2, 1>},
                                        {APT_FLAT_FN(k_extend_flat<0, 0>), APT_FLAT_FN(k_extend_flat<1, 0>)}};   // [mode][unsorted | sorted into the packed class queues] (flat: the self-contained variant, for explicit rays)
// flat sweep inside a render: the hot variant and its fix-up launch (stages.hpp "fix-up lists"), [sorted]
static const extend_fn kExtendFlatHot[2] = {APT_FLAT_FN(k_extend_flat<0, 1>), APT_FLAT_FN(k_extend_flat<1, 1>)};
static const extend_fn kFixFlat[2] = {APT_FLAT_FN(k_fix_flat<0>), APT_FLAT_FN(k_fix_flat<1>)};
static const extend_fn kExtendDyn[2] = {k_extend_dyn<0>, k_extend_dyn<1>};      // BVH walk with dynamic ray fetch [sorted]
static const shadow_fn kShadow[4] = {k_shadow<0>, k_shadow<1>, k_shadow<2>, APT_FLAT_FN(k_shadow_flat<1>)};      // (flat: the hot variant; its list is served by the next kFixFlat launch)
static const occluded_fn kOccluded[4] = {k_occluded<0>, k_occluded<1>, k_occluded<2>, APT_FLAT_FN(k_occluded_flat)};
// Volumetric shading sorted by EVENT (volumetric.hpp k_vevent / k_vshade_ev): one queue per surface class (kClassMask order; the volumetric
// tracer does not split Blinn-Phong by lobe) and one for the medium, shaded by the group kernels below; textured scenes, or scenes with
// more classes than queues, keep ONE surface queue, shaded by the all-models kernel: [emitter set: point + area | all][without / with a grid volume]
typedef void (*vevent_fn)(DevScene, Params, Queues, Counters*, int, int);
typedef void (*vev_shade_fn)(DevScene, Params, Queues, Counters*, int, int);
static const vevent_fn kVEvent[2] = {k_vevent<0>, k_vevent<1>};
static const vev_shade_fn kVEventAll[2][2] = {{k_vshade_ev<APT_BX_ALL, 0x03, 0, 0>, k_vshade_ev<APT_BX_ALL, 0x03, 1, 0>}, {k_vshade_ev<APT_BX_ALL, APT_SRC_ALL, 0, 0>, k_vshade_ev<APT_BX_ALL, APT_SRC_ALL, 1, 0>}};
// ... launched in register-footprint groups (volumetric.hpp k_vshade_ev_group): class definition -> (group, member slot); the medium is member 0 of group 1
typedef void (*vev_group_fn)(DevScene, Params, Queues, Counters*, VGroupIn, int);
#define APT_N_VGROUPS 3
static const int kVClassGroup[APT_N_CLASS_DEFS] = {0, 2, 0, 0, 2, 2, 1, 0, 1, 2};
static const int kVClassSlot[APT_N_CLASS_DEFS] = {0, 0, 1, 2, 1, 2, 1, 3, 2, 3};
#define APT_VGROUP_ROW(SM, VOL) {k_vshade_ev_group<SM, VOL, 4, 0x002, 0x040, 0x504, 0x200>, k_vshade_ev_group<SM, VOL, 3, APT_VEV_MEDIUM_CODE, 0x080, 0x008, 0>, k_vshade_ev_group<SM, VOL, 1, 0x001, 0x010, 0x020, 0x001>}
static const vev_group_fn kVGroup[2][2][APT_N_VGROUPS] = {{APT_VGROUP_ROW(0x03, 0), APT_VGROUP_ROW(0x03, 1)}, {APT_VGROUP_ROW(APT_SRC_ALL, 0), APT_VGROUP_ROW(APT_SRC_ALL, 1)}};      // [emitter set][grid volume][group]
typedef void (*vshadow_fn)(DevScene, Params, Queues, Counters*, LdsPlan, int);
static const vshadow_fn kVShadow[5] = {k_vshadow<0>, k_vshadow<1>, k_vshadow<2>, nullptr, APT_FLAT_FN(k_vshadow_flat)};     // volumetric: transmittance walk (one closest-hit query per pass; [4]: the flat sweep, two samples per lane, every segment in one launch)
#define APT_SWEEP_MAX_PRIMS 96   // up to here the uniform sweep beats the BVH walk (no divergence, scalar loads)

// ============================================================== host side
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { if (p) (void)hipFree(p); p = nullptr; bytes = n; return hipMalloc(&p, n ? n : 4); }
    template <class T> T* as() const { return (T*)p; }
};
template <class T> static hipError_t upload(DevBuf& b, const std::vector<T>& v) {
    hipError_t e = b.alloc(v.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return v.empty() ? hipSuccess : hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

struct apt_bvh { apt::BvhData data; apt::BvhData bin3; apt::WideBvhData wide; };     // data: the exported binary tree (leaves <= 4); bin3 (single-primitive leaves) -> wide: what the kernels walk

struct apt_scene {
    int device = 0;
    DevScene dev{};
    apt::BvhData bvh;                    // binary SAH tree (leaves of <= 3 primitives): the intermediate of the build
    apt::WideBvhData wide;               // 8-wide quantised tree: what the kernels walk
    DevBuf nodes, prims, slot_prim, normals, vnormals, precom, prim_obj, prim_class, obj_info, emitter_id, bxdf, src, sweep_recs, sweep_tab, obj_aabb;
    DevBuf flat_pairs;                   // the flat records two by two (traverse.hpp FlatScene::pairs)
    DevBuf flat_recs, flat_tab;          // flat sweep (fast build, small scenes): records and the per-record table (traverse.hpp FlatScene)
    bool has_flat = false;
    DevBuf uvs, tex_i, tex_f, atlas[3];      // image textures (empty when the scene has none)
    DevBuf prim_shade;                   // per-primitive shading records (stages.hpp DevScene::prim_shade)
    DevBuf med;                          // participating media, n_objects + 1 rows (volumetric path tracer)
    DevBuf vol_grid;                     // grid volume densities
    bool has_volume = false;
    bool gpu_built = false;              // the binary tree came from the device builder (bvh_gpu.hip)
    bool world_scattering = false;       // the world medium scatters (rays that hit nothing still take part, vpt.py:176-181)
    bool has_null_surface = false;       // some object carries a null BSDF (rays pass, vpt.py:189-191)
    std::vector<int> obj_class;          // per object: compact material class
    std::vector<uint8_t> obj_null;       // per object: null BSDF (never shaded)
    bool phong_no_lobe = true;           // every Blinn-Phong material has k_s = 0 and finite k_g >= 0
    float box_min[3] = {1e3f, 1e3f, 1e3f}, box_max[3] = {-1e3f, -1e3f, -1e3f};    // union of the object boxes (path_tracer.py:130-134)
    int n_classes = 0;                   // material classes present (compact ids 0..n_classes-1)
    int class_def[APT_N_CLASS_DEFS] = {};   // compact id -> class definition
    int n_prims = 0, n_objects = 0, n_sources = 0, max_obj_prims = 0;
    int bx_mask = 0, src_mask = 0;
    bool has_aabb = false;
    bool has_sweep = false;              // the sweep stream exists (scenes small enough that a sweep could ever be asked for)
};

struct EventPair { hipEvent_t a, b; int kernel; bool own_a; };       // own_a = false: `a` is the previous launch's `b` (back-to-back launches on one stream share the event)

struct apt_renderer {
    const apt_scene* scene = nullptr;
    apt_render_cfg cfg{};
    Params par{};
    Queues q{};
    int n_cols = 0, npix = 0, spp_batch = 1;
    int cnt = 0;
    hipStream_t stream = nullptr;
    DevBuf pool, counters, accum, scratch, pix_key;
    Counters host_counters{};
    int grid_small = 0, grid_trace = 0, nq = APT_MAX_NQ;
    const ShadeVariant* shade = nullptr;
    int trace_mode = 0;           // 0 = BVH traversal, 1 = wave-uniform sweep, 2 = tiled sweep (small scenes)
    int trace_nt = BLOCK;         // workgroup size of the trace kernels
    int trace_items = BLOCK;      // queue entries per workgroup pass (flat sweep: two per thread)
    int sorted = 0;               // 1 = material-sorted shading (>= 2 material classes in the scene)
    int volumetric = 0;           // 1 = VolumeRenderer.render semantics (volumetric.hpp)
    int dyn_fetch = 0;            // BVH mode: closest-hit walk with dynamic ray fetch (k_extend_dyn)
    int ovf_levels = 0;           // traversal-stack levels beyond the LDS part
    int v_ncls = 0;               // class queues in use (surface classes + the miss class when misses matter)
    int vevent = 0;               // volumetric shading sorted by event: k_vevent decides what every path does, one kernel per event queue (v_ncls of them: surface classes, the medium last)
    int vev_single = 0;           // ... with ONE surface queue served by the all-models kernel (textured scenes, more classes than queues)
    vev_shade_fn vev_all = nullptr;
    bool vev_live[APT_MAX_CLASSES] = {};      // an event queue that can receive entries at all (a class of null surfaces only is never shaded)
    vev_group_fn vgroup_fn[APT_N_VGROUPS] = {};           // ... the event kernels in groups
    int vgroup_cls[APT_N_VGROUPS][4] = {};                // ... event queue of each member slot (-1: none)
    group_fn group_fn_[APT_N_GROUPS] = {};                // ... the group kernels for this scene's emitter set
    int group_cls[APT_N_GROUPS][APT_GROUP_SLOTS] = {};    // ... compact class id of each member slot (-1: the scene has no such class)
    std::string shade_name;
    LdsPlan plan{};
    size_t lds_bytes = 0, lds_bytes_any = 0;     // dynamic LDS of the closest-hit / any-hit trace kernels
    int grid_shadow = 0;
    int grid_fix = 0;             // flat sweep: grid of the fix-up launches (a few workgroups per sub-queue: their lists are all but empty)
    int grid_vshadow = 0;         // volumetric transmittance walk (closest-hit LDS footprint, its own register budget)
    int vshadow_nt = BLOCK;       // its workgroup size and dynamic LDS
    int vshadow_mode = 0;         // traversal mode of the volumetric transmittance walk (one closest-hit query per lane and pass: with the flat sweep's two-rays-per-lane loop half of every packed instruction would idle, so small scenes keep the tiled / wave sweep there)
    size_t vshadow_lds = 0;
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> free_events;
    std::vector<std::pair<hipStream_t, hipEvent_t>> chain;      // per stream: the end event of a timed launch nothing has followed yet
    double kernel_ms[APT_N_KERNELS] = {0, 0, 0, 0, 0};
    int64_t launches[APT_N_KERNELS] = {0, 0, 0, 0, 0};
    double render_ms = 0.0;
    hipEvent_t ev_r0 = nullptr, ev_r1 = nullptr;
    bool render_pending = false;
    // Render lanes: independent batch pipelines (own stream, queue pool, counters) whose kernels overlap on the GPU —
    // one lane's latency-bound shade runs beside another lane's VALU-bound extend.  Lane 0 is {stream, pool, counters, q}
    // above; the framebuffer is shared and the finalize kernels are chained by events so that samples are still added to
    // a pixel in sample order (bit-identical to the single-lane result).
    struct Lane { hipStream_t stream = nullptr; DevBuf pool, counters, ovf; Queues q{}; hipEvent_t fin = nullptr; LdsPlan plan{}; };
    DevBuf ovf;                   // lane 0: global spill columns of the BVH traversal stack
    std::vector<Lane> extra;      // lanes 1..n-1
    hipEvent_t fin0 = nullptr;    // lane 0's "finalize done" event
    int n_lanes = 1;
};

static int count_device(int* n) {
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess || *n <= 0) { *n = 0; return fail(APT_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)"); }
    return APT_OK;
}

APT_EXPORT const char* apt_last_error(void) { return g_err.c_str(); }
#if APT_FAST
APT_EXPORT const char* apt_version(void) { return "adapt_mi 0.3 (gfx950 wavefront path tracer; arithmetic: fast)"; }
#else
APT_EXPORT const char* apt_version(void) { return "adapt_mi 0.3 (gfx950 wavefront path tracer; arithmetic: exact)"; }
#endif

// ---- BVH build (host only, no device needed)
APT_EXPORT int apt_bvh_build(const float* prims, int32_t n_prims, const int32_t* obj_info, int32_t n_objects, apt_bvh** out) {
    if (!prims || !obj_info || !out || n_prims <= 0 || n_objects <= 0) return fail(APT_E_INVALID, "apt_bvh_build: bad argument");
    apt_bvh* b = new apt_bvh();
    if (apt::build_bvh(prims, n_prims, obj_info, n_objects, b->data) != 0 || apt::build_bvh(prims, n_prims, obj_info, n_objects, b->bin3, 1) != 0 ||
        apt::build_wide_bvh(b->bin3, b->wide) != 0) { delete b; return fail(APT_E_INVALID, "apt_bvh_build: build failed"); }
    *out = b;
    return APT_OK;
}
APT_EXPORT int apt_bvh_counts(const apt_bvh* b, int32_t* n_nodes, int32_t* n_leaf_prims, int32_t* max_depth) {
    if (!b) return fail(APT_E_INVALID, "apt_bvh_counts: null handle");
    if (n_nodes) *n_nodes = b->data.n_nodes();
    if (n_leaf_prims) *n_leaf_prims = (int32_t)b->data.prim_order.size();
    if (max_depth) *max_depth = b->data.max_depth;
    return APT_OK;
}
APT_EXPORT int apt_bvh_export(const apt_bvh* b, float* nodes, int32_t* prim_order) {
    if (!b || !nodes || !prim_order) return fail(APT_E_INVALID, "apt_bvh_export: bad argument");
    memcpy(nodes, b->data.nodes.data(), b->data.nodes.size() * sizeof(float));
    memcpy(prim_order, b->data.prim_order.data(), b->data.prim_order.size() * sizeof(int32_t));
    return APT_OK;
}
APT_EXPORT int apt_flat_records(const float* prims, int32_t n_prims, const int32_t* obj_info, int32_t n_objects, int32_t counts[7],
                                float* stream, int32_t stream_cap, float* tab, int32_t tab_cap, int32_t* n_stream, int32_t* n_tab) {
    if (!prims || !obj_info || !counts || n_prims <= 0 || n_objects <= 0 || !n_stream || !n_tab) return fail(APT_E_INVALID, "apt_flat_records: bad argument");
    std::vector<float> st, tb; int c[7];
    if (apt::build_flat(prims, n_prims, obj_info, n_objects, nullptr, nullptr, st, tb, c) != 0) return fail(APT_E_INVALID, "apt_flat_records: obj_info range outside the primitive array");
    for (int k = 0; k < 7; k++) counts[k] = c[k];
    *n_stream = (int32_t)st.size(); *n_tab = (int32_t)tb.size();
    if (stream) { if (stream_cap < (int32_t)st.size()) return fail(APT_E_INVALID, "apt_flat_records: stream buffer too small"); memcpy(stream, st.data(), st.size() * 4); }
    if (tab) { if (tab_cap < (int32_t)tb.size()) return fail(APT_E_INVALID, "apt_flat_records: table buffer too small"); memcpy(tab, tb.data(), tb.size() * 4); }
    return APT_OK;
}
APT_EXPORT int apt_bvh_wide_counts(const apt_bvh* b, int32_t* n_nodes, int32_t* n_levels) {
    if (!b) return fail(APT_E_INVALID, "apt_bvh_wide_counts: null handle");
    if (n_nodes) *n_nodes = b->wide.n_nodes();
    if (n_levels) *n_levels = b->wide.max_depth;
    return APT_OK;
}
APT_EXPORT int apt_bvh_wide_export(const apt_bvh* b, uint32_t* nodes, int32_t* prim_order) {
    if (!b || !nodes || !prim_order) return fail(APT_E_INVALID, "apt_bvh_wide_export: bad argument");
    memcpy(nodes, b->wide.nodes.data(), b->wide.nodes.size() * sizeof(uint32_t));
    memcpy(prim_order, b->wide.prim_order.data(), b->wide.prim_order.size() * sizeof(int32_t));
    return APT_OK;
}
APT_EXPORT int apt_bvh_wide_frame(const apt_bvh* b, float gmin[3], float gstep[3]) {
    if (!b || !gmin || !gstep) return fail(APT_E_INVALID, "apt_bvh_wide_frame: bad argument");
    for (int a = 0; a < 3; a++) { gmin[a] = b->wide.frame.gmin[a]; gstep[a] = b->wide.frame.gstep[a]; }
    return APT_OK;
}
APT_EXPORT void apt_bvh_free(apt_bvh* b) { delete b; }

// ---- `bvh_cpp.bvh_build`-compatible export (host only): the reference-layout tree for AdaPT's own traversal kernels
struct apt_linear_bvh { apt::LinearBvhData data; };
APT_EXPORT int apt_bvh_build_linear(const float* prims, int32_t n_prims, const int32_t* obj_prim_cnt, const int32_t* obj_is_sphere, int32_t n_objects,
                                    const float* world_min, const float* world_max, apt_linear_bvh** out) {
    if (!prims || !obj_prim_cnt || !obj_is_sphere || !world_min || !world_max || !out || n_prims <= 0 || n_objects <= 0)
        return fail(APT_E_INVALID, "apt_bvh_build_linear: bad argument");
    apt_linear_bvh* b = new apt_linear_bvh();
    if (apt::build_linear_bvh(prims, n_prims, obj_prim_cnt, obj_is_sphere, n_objects, world_min, world_max, b->data) != 0) {
        delete b;
        return fail(APT_E_INVALID, "apt_bvh_build_linear: the per-object primitive counts must be non-negative and sum to n_prims");
    }
    *out = b;
    return APT_OK;
}
APT_EXPORT int apt_linear_bvh_counts(const apt_linear_bvh* b, int32_t* n_nodes, int32_t* n_prims) {
    if (!b) return fail(APT_E_INVALID, "apt_linear_bvh_counts: null handle");
    if (n_nodes) *n_nodes = b->data.n_nodes();
    if (n_prims) *n_prims = b->data.n_prims();
    return APT_OK;
}
APT_EXPORT int apt_linear_bvh_export(const apt_linear_bvh* b, float* bvh_minmax, float* node_minmax, int32_t* bvh_info, int32_t* node_info) {
    if (!b || !bvh_minmax || !node_minmax || !bvh_info || !node_info) return fail(APT_E_INVALID, "apt_linear_bvh_export: bad argument");
    memcpy(bvh_minmax, b->data.bvh_minmax.data(), b->data.bvh_minmax.size() * sizeof(float));
    memcpy(node_minmax, b->data.node_minmax.data(), b->data.node_minmax.size() * sizeof(float));
    memcpy(bvh_info, b->data.bvh_info.data(), b->data.bvh_info.size() * sizeof(int32_t));
    memcpy(node_info, b->data.node_info.data(), b->data.node_info.size() * sizeof(int32_t));
    return APT_OK;
}
APT_EXPORT void apt_linear_bvh_free(apt_linear_bvh* b) { delete b; }

// ---- scene
APT_EXPORT int apt_scene_create(const apt_scene_desc* d, int32_t device, apt_scene** out) {
    if (!d || !out) return fail(APT_E_INVALID, "apt_scene_create: null argument");
    if (d->n_prims <= 0 || d->n_objects <= 0 || d->n_sources <= 0 || !d->prims || !d->normals || !d->obj_info || !d->emitter_id ||
        !d->bxdf_i || !d->bxdf_f || !d->src_i || !d->src_f)
        return fail(APT_E_INVALID, "apt_scene_create: incomplete scene description");
    int ndev = 0;
    if (int rc = count_device(&ndev)) return rc;
    if (device < 0 || device >= ndev) return fail(APT_E_INVALID, "apt_scene_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    apt_scene* s = new apt_scene();
    s->device = device;
    const int N = d->n_prims, O = d->n_objects, S = d->n_sources;
    // the 64-byte nodes of the walk hold child and primitive indices in 24 bits (bvh_wide.cpp); records are addressed with 32-bit byte offsets
    if (N >= (1 << 24)) { delete s; return fail(APT_E_INVALID, "apt_scene_create: more than 16 777 215 primitives (24-bit indices in the 64-byte tree nodes)"); }
    s->n_prims = N; s->n_objects = O; s->n_sources = S;
    const bool timing = getenv("APT_SCENE_TIMING") != nullptr;      // stderr: where apt_scene_create spends its time
    auto t_prev = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[scene timing] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    const int max_leaf = 1;               // primitives per leaf of the binary tree: the 64-byte node's leaf child IS one primitive (measured 1 / 2 / 3 per leaf on the 80-byte node, rounds 2 and 5: C4 1274 / 1236 / 1228, C5 1117 / 1073 / 1043 Msamples/s; C4 extend 34.1 / 33.5 / 33.5 ms, C5's any-hit walk 25.7 / 27.9 / 28.9)
    // builder: binned SAH on the host (best tree) below a million primitives, PLOC on the device above (scene-load time); APT_BVH_BUILDER=sah|ploc|lbvh overrides.
    // Measured on one MI355X (Msamples/s, C4 95 k / C5 285 k triangles): SAH 1303 / 1216, PLOC 1287 / 1157, LBVH 1186 / 1048; apt_scene_create at 1.14 M
    // primitives: SAH ~800 ms, PLOC or LBVH ~500 ms (what is left is the 8-wide collapse and the table uploads, shared by all three)
    bool gpu_build = N >= 1000000;
    int gpu_algo = 1;                     // 0 LBVH (radix tree), 1 PLOC (nearest-neighbour merging by box area)
    if (const char* bb = getenv("APT_BVH_BUILDER")) { gpu_build = (!strcmp(bb, "lbvh") || !strcmp(bb, "ploc")) && N >= 2; gpu_algo = !strcmp(bb, "ploc") ? 1 : 0; }
    s->gpu_built = false;
    if (gpu_build) {
        const int rc_ = apt::build_bvh_gpu(d->prims, N, d->obj_info, O, device, s->bvh, gpu_algo);
        if (rc_ != 0) {                                      // a valid scene must load: the host builder takes over (slower, never fails on valid input)
            fprintf(stderr, "adapt_mi: device BVH build failed (%d), falling back to the host SAH builder\n", rc_);
            if (apt::build_bvh(d->prims, N, d->obj_info, O, s->bvh, max_leaf) != 0) { delete s; return fail(APT_E_INVALID, "apt_scene_create: BVH build failed"); }
            gpu_build = false;
        } else s->gpu_built = true;
    } else if (apt::build_bvh(d->prims, N, d->obj_info, O, s->bvh, max_leaf) != 0) { delete s; return fail(APT_E_INVALID, "apt_scene_create: BVH build failed"); }
    tick(gpu_build ? (gpu_algo ? "binary tree (PLOC, device)" : "binary tree (LBVH, device)") : "binary tree (SAH, host)");
    if (apt::build_wide_bvh(s->bvh, s->wide) != 0) { delete s; return fail(APT_E_INVALID, "apt_scene_create: BVH collapse failed"); }
    tick("8-wide collapse");

    std::vector<int> prim_obj((size_t)N, 0);
    std::vector<uint8_t> sphere((size_t)N, 0);
    for (int o = 0; o < O; o++)
        for (int k = d->obj_info[3 * o]; k < d->obj_info[3 * o] + d->obj_info[3 * o + 1]; k++) {
            if (k < 0 || k >= N) { delete s; return fail(APT_E_INVALID, "apt_scene_create: obj_info range outside the primitive array"); }
            prim_obj[(size_t)k] = o; sphere[(size_t)k] = d->obj_info[3 * o + 2] != 0;
        }
    // primitive records in BVH order + precom rows (tracer_base.py:117-134)
    std::vector<float> prec((size_t)N * 9), recs((size_t)N * 12, 0.f);
    const int host_thr = apt::host_threads();           // per-primitive loops below: independent rows, chunked over host threads (bvh_build.hpp)
    apt::parallel_for(N, host_thr, [&](int k) {
        const float* v = d->prims + 9 * (size_t)k; float* pc = prec.data() + 9 * (size_t)k;
        if (sphere[(size_t)k]) { for (int a = 0; a < 6; a++) pc[a] = v[a]; for (int a = 0; a < 3; a++) pc[6 + a] = v[a]; }
        else for (int a = 0; a < 3; a++) { pc[a] = v[3 + a] - v[a]; pc[3 + a] = v[6 + a] - v[a]; pc[6 + a] = v[a]; }
    });
    apt::parallel_for(N, host_thr, [&](int slot) {
        int k = s->wide.prim_order[(size_t)slot];
        const float* v = d->prims + 9 * (size_t)k; const float* pc = prec.data() + 9 * (size_t)k; float* r = recs.data() + 12 * (size_t)slot;
        int32_t kid = k, flag = sphere[(size_t)k] ? 1 : 0;
#if APT_FAST_LEAVES
        // product build: precomputed-transform records without an id (traverse.hpp tri_two; the walk reports leaf slots, DevBvh::slot_prim)
        (void)pc; (void)kid;
        if (flag) { r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = std::numeric_limits<float>::quiet_NaN(); r[4] = v[3]; }
        else apt::planar_rows(v, r);
#else
        if (flag) { r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; }
        else { r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = pc[0]; r[4] = pc[1]; r[5] = pc[2]; r[6] = pc[3]; r[7] = pc[4]; r[8] = pc[5]; }
        memcpy(&r[9], &kid, 4); memcpy(&r[10], &flag, 4);
#endif
    });
    tick("primitive records");
    // sweep stream in scene order (layout: traverse.hpp SweepScene); the t-row cofactors of [e1 e2 .] are
    // ray-independent, so they are computed once here with the same float operations the device would use
    std::vector<float> sw;
    std::vector<int> sw_tab((size_t)O * 4, 0);
    s->has_sweep = N < 65536;            // larger scenes always walk the BVH (and would pay 48 B per primitive and their share of the load time for a stream nobody reads)
    for (int o = 0; o < O; o++) {
        const int first = d->obj_info[3 * o], count = d->obj_info[3 * o + 1], is_sphere = d->obj_info[3 * o + 2] != 0;
        if (!is_sphere) s->max_obj_prims = std::max(s->max_obj_prims, count);
        if (!s->has_sweep) continue;
        sw_tab[4 * (size_t)o] = (int)sw.size(); sw_tab[4 * (size_t)o + 1] = count; sw_tab[4 * (size_t)o + 2] = is_sphere; sw_tab[4 * (size_t)o + 3] = first;
        size_t base = sw.size();
        sw.resize(base + 8, 0.f);
        if (d->obj_aabb) for (int a = 0; a < 3; a++) { sw[base + 2 * a] = d->obj_aabb[6 * o + a]; sw[base + 2 * a + 1] = d->obj_aabb[6 * o + 3 + a]; }
        if (is_sphere) {
            size_t at = sw.size(); sw.resize(at + 8, 0.f);
            for (int a = 0; a < 4; a++) sw[at + a] = d->prims[9 * (size_t)first + a];
            continue;
        }
        const int n_pairs = (count + 1) / 2;
        size_t at = sw.size(); sw.resize(at + (size_t)n_pairs * 24, 0.f);
        for (int k = 0; k < count; k++) {
            const float* v = d->prims + 9 * (size_t)(first + k); const float* pc = prec.data() + 9 * (size_t)(first + k);
            float* r = sw.data() + at + 24 * (size_t)(k / 2) + (k & 1);
            const float a00 = pc[0], a10 = pc[1], a20 = pc[2], a01 = pc[3], a11 = pc[4], a21 = pc[5];
            for (int a = 0; a < 3; a++) r[2 * a] = v[a];
            for (int a = 0; a < 6; a++) r[6 + 2 * a] = pc[a];
            r[18] = a10 * a21 - a20 * a11; r[20] = a20 * a01 - a00 * a21; r[22] = a00 * a11 - a10 * a01;
        }
    }
    tick("sweep stream");
    std::vector<float> aabb((size_t)O * 6, 0.f);
    if (d->obj_aabb) aabb.assign(d->obj_aabb, d->obj_aabb + (size_t)O * 6);
    std::vector<DevBxdf> bx((size_t)O);
    std::vector<uint8_t> no_lobe((size_t)O, 0);
    for (int o = 0; o < O; o++) {
        const int32_t* bi = d->bxdf_i + 4 * o; const float* bf = d->bxdf_f + 13 * o;
        DevBxdf& b = bx[(size_t)o]; memset(&b, 0, sizeof(b));
        b.type = bi[0]; b.is_delta = bi[1]; b.is_bsdf = bi[2];
        s->bx_mask |= b.is_bsdf ? (b.type == 0 ? 0x100 : (b.type == 1 ? 0x200 : 0x400)) : (1 << (b.type & 7));
        b.k_d = mk3(bf[0], bf[1], bf[2]); b.k_s = mk3(bf[3], bf[4], bf[5]); b.k_g = mk3(bf[6], bf[7], bf[8]); b.mean = mk3(bf[9], bf[10], bf[11]); b.ior = bf[12];
        no_lobe[(size_t)o] = 0;
        if (!b.is_bsdf && b.type == 0) {
            bool lean = true;
            for (int a = 0; a < 3; a++) if (!(bf[3 + a] == 0.f && !std::signbit(bf[3 + a]) && bf[6 + a] >= 0.f && std::isfinite(bf[6 + a]))) lean = false;
            no_lobe[(size_t)o] = lean ? 1 : 0;
            if (!lean) s->phong_no_lobe = false;
        }
    }
    std::vector<int> pcls_host;
    // material classes present in this scene -> compact ids; per-primitive class table for the sorting extend.  Blinn-Phong objects
    // without a specular lobe (the diffuse walls of most scenes) get a class of their own, whose kernel carries no double-precision
    // pow - unless that would need more class queues than there are (APT_MAX_CLASSES), then they stay with the other Blinn-Phong objects
    {
        int compact[APT_N_CLASS_DEFS];
        std::vector<int> obj_cls((size_t)O);
        for (int split = 1; split >= 0; split--) {
            for (int c = 0; c < APT_N_CLASS_DEFS; c++) compact[c] = -1;
            s->n_classes = 0;
            for (int o = 0; o < O; o++) {
                int c = class_of(bx[(size_t)o].is_bsdf, bx[(size_t)o].type, split && no_lobe[(size_t)o]);
                if (compact[c] < 0) { compact[c] = s->n_classes; s->class_def[s->n_classes++] = c; }
                obj_cls[(size_t)o] = compact[c];
            }
            if (s->n_classes <= APT_MAX_CLASSES) break;
        }
        s->obj_class = obj_cls;
        s->obj_null.resize((size_t)O);
        for (int o = 0; o < O; o++) s->obj_null[(size_t)o] = (bx[(size_t)o].is_bsdf && bx[(size_t)o].type < 0) ? 1 : 0;
        std::vector<int> pcls((size_t)N);
        for (int k = 0; k < N; k++) pcls[(size_t)k] = obj_cls[(size_t)prim_obj[(size_t)k]];
        pcls_host = pcls;
        hipError_t e_ = upload(s->prim_class, pcls);
        if (e_ != hipSuccess) { delete s; return fail(APT_E_HIP, std::string("upload prim_class: ") + hipGetErrorString(e_)); }
        memset(&s->dev.flat, 0, sizeof(s->dev.flat));
#if APT_FAST
        if (N <= APT_FLAT_MAX_PRIMS && s->has_sweep && d->obj_aabb) {          // flat sweep records (traverse.hpp "Flat sweep"); the sweep stream serves its zero-component fallback
            std::vector<float> fr, ft;
            FlatScene& fl = s->dev.flat;
            std::vector<uint8_t> trans((size_t)O);
            for (int o = 0; o < O; o++) trans[(size_t)o] = bx[(size_t)o].is_bsdf ? 1 : 0;
            int fc[7];
            if (apt::build_flat(d->prims, N, d->obj_info, O, pcls.data(), trans.data(), fr, ft, fc) != 0) { delete s; return fail(APT_E_INVALID, "apt_scene_create: flat records: obj_info range outside the primitive array"); }
            fl.defer_all = getenv("APT_FLAT_DEFER_ALL") ? atoi(getenv("APT_FLAT_DEFER_ALL")) : 0;
            fl.n_quads = fc[0]; fl.n_quads_tie = fc[1]; fl.n_gquads = fc[2]; fl.n_gquads_tie = fc[3]; fl.n_tris = fc[4]; fl.n_tris_tie = fc[5]; fl.n_spheres = fc[6];
            hipError_t e1_ = upload(s->flat_recs, fr), e2_ = (e1_ == hipSuccess) ? upload(s->flat_tab, ft) : e1_;
            if (e2_ != hipSuccess) { delete s; return fail(APT_E_HIP, std::string("upload flat records: ") + hipGetErrorString(e2_)); }
            fl.stream = s->flat_recs.as<float>(); fl.tab = s->flat_tab.as<float4>();
            {   // the records two by two for the one-ray any-hit sweep (flat_any1): floats of records 2j, 2j + 1 interleaved; an odd tail repeats its record
                std::vector<float> fp;
                const float* src = fr.data();
                auto section = [&](int n, int w) {
                    for (int j = 0; 2 * j < n; j++) {
                        const float* a = src + (size_t)(2 * j) * w; const float* b = (2 * j + 1 < n) ? a + w : a;
                        for (int k = 0; k < w; k++) { fp.push_back(a[k]); fp.push_back(b[k]); }
                    }
                    src += (size_t)n * w;
                };
                section(fc[0] + fc[1], 12); section(fc[2] + fc[3], 18); section(fc[4] + fc[5], 12); section(fc[6], 4);
                if (fp.empty()) fp.push_back(0.f);
                hipError_t e3_ = upload(s->flat_pairs, fp);
                if (e3_ != hipSuccess) { delete s; return fail(APT_E_HIP, std::string("upload flat record pairs: ") + hipGetErrorString(e3_)); }
                fl.pairs = s->flat_pairs.as<float>();
            }
            s->has_flat = true;
            if (timing) fprintf(stderr, "[scene timing] flat records: %d + %d parallelograms, %d + %d convex quads, %d + %d triangles (plain + coplanar groups), %d spheres of %d primitives\n", fc[0], fc[1], fc[2], fc[3], fc[4], fc[5], fc[6], N);
            tick("flat records");
        }
#endif
    }
    std::vector<DevSrc> sr((size_t)S);
    for (int k = 0; k < S; k++) {
        const int32_t* si = d->src_i + 4 * k; const float* sf = d->src_f + 11 * k;
        DevSrc& e = sr[(size_t)k]; memset(&e, 0, sizeof(e));
        e.type = si[0]; e.bool_bits = si[1]; e.obj_ref_id = si[2];
        s->src_mask |= 1 << (e.type & 7);
        e.intensity = mk3(sf[0], sf[1], sf[2]); e.dir = mk3(sf[3], sf[4], sf[5]); e.pos = mk3(sf[6], sf[7], sf[8]); e.inv_area = sf[9]; e.r = sf[10];
        if (e.type == 1 && (e.obj_ref_id < 0 || e.obj_ref_id >= O)) { delete s; return fail(APT_E_INVALID, "apt_scene_create: area emitter is not attached to an object"); }
        if (e.type == 1) { e.prim_first = d->obj_info[3 * e.obj_ref_id]; e.prim_count = d->obj_info[3 * e.obj_ref_id + 2] ? -1 : d->obj_info[3 * e.obj_ref_id + 1]; }
    }
    std::vector<float> nrm(d->normals, d->normals + (size_t)N * 3);
    std::vector<float> vn((size_t)N * 12, 0.f);           // three 16-byte records per primitive (DevScene::vnormals)
    if (d->v_normals) for (size_t k = 0; k < (size_t)N * 3; k++) for (int a = 0; a < 3; a++) vn[4 * k + a] = d->v_normals[3 * k + a];
    std::vector<int> oi(d->obj_info, d->obj_info + (size_t)O * 3), ei(d->emitter_id, d->emitter_id + (size_t)O);
#define UP(buf, vec) do { hipError_t e_ = upload(s->buf, vec); if (e_ != hipSuccess) { delete s; return fail(APT_E_HIP, std::string("upload " #buf ": ") + hipGetErrorString(e_)); } } while (0)
    std::vector<int> slot_info((size_t)N);                // leaf slot -> primitive | material class << 28 (traverse.hpp walk_info)
    {
        // Three class bits (28..30), so that the word stays non-negative: k_extend_dyn reads a negative word as "nothing hit".  A scene
        // with more classes than class queues (all nine surface models + the lobe-free walls) renders unsorted through the all-models
        // kernel, nobody reads the class then, and none is packed (a compact id of 8 would have set the sign bit: every closest hit on
        // the ninth class lost).
        if (N >= (1 << 28)) { delete s; return fail(APT_E_INVALID, "apt_scene_create: more than 2^28 primitives"); }
        const bool pack_cls = s->n_classes <= APT_MAX_CLASSES;
        for (int slot = 0; slot < N; slot++) {
            const int k = s->wide.prim_order[(size_t)slot];
            const uint32_t c = pack_cls ? (uint32_t)pcls_host[(size_t)k] : 0u;
            if (c >= (uint32_t)APT_MAX_CLASSES) { delete s; return fail(APT_E_INVALID, "apt_scene_create: material class id does not fit the leaf-slot word"); }
            slot_info[(size_t)slot] = (int)((uint32_t)k | (c << 28));
        }
    }
    UP(nodes, s->wide.nodes); UP(prims, recs); UP(slot_prim, slot_info); UP(normals, nrm); UP(vnormals, vn); UP(precom, prec); UP(prim_obj, prim_obj);
    UP(obj_info, oi); UP(emitter_id, ei); UP(bxdf, bx); UP(src, sr); UP(sweep_recs, sw); UP(sweep_tab, sw_tab); UP(obj_aabb, aabb);
    DevScene& ds = s->dev;
    ds.bvh.nodes = s->nodes.as<uint4>(); ds.bvh.prims = s->prims.as<float4>(); ds.bvh.slot_prim = s->slot_prim.as<int>(); ds.bvh.n_nodes = s->wide.n_nodes(); ds.bvh.n_prims = N;
    for (int a = 0; a < 3; a++) { ds.bvh.gmin[a] = s->wide.frame.gmin[a]; ds.bvh.gstep[a] = s->wide.frame.gstep[a]; ds.bvh.ginv[a] = 1.0f / s->wide.frame.gstep[a]; }
    ds.sweep.stream = s->sweep_recs.as<float>(); ds.sweep.obj_tab = s->sweep_tab.as<int>(); ds.sweep.prim_obj = s->prim_obj.as<int>(); ds.sweep.n_objects = O;
    s->has_aabb = d->obj_aabb != nullptr;
    ds.normals = s->normals.as<float>(); ds.vnormals = s->vnormals.as<float4>(); ds.precom = s->precom.as<float>();
    ds.flat.precom = ds.precom;
    ds.prim_obj = s->prim_obj.as<int>(); ds.prim_class = s->prim_class.as<int>(); ds.obj_info = s->obj_info.as<int>(); ds.emitter_id = s->emitter_id.as<int>();
    ds.bxdf = s->bxdf.as<DevBxdf>(); ds.src = s->src.as<DevSrc>();
    ds.n_prims = N; ds.n_objects = O; ds.n_sources = S; ds.has_vn = d->has_vertex_normal; ds.world_ior = d->world_ior;
    ds.uvs = nullptr; ds.tex_i = nullptr; ds.tex_f = nullptr;
    for (int m = 0; m < 3; m++) { ds.atlas[m] = nullptr; ds.atlas_w[m] = 0; }
    if (d->tex_i && d->tex_f && d->uvs) {
        for (int o = 0; o < O; o++) for (int m = 0; m < 3; m++) {
            const int32_t* t = d->tex_i + 15 * o + 5 * m;
            if (t[0] <= -255) continue;
            if (d->obj_info[3 * o + 2]) { delete s; return fail(APT_E_INVALID, "apt_scene_create: textured spheres are not supported"); }
            if (!d->atlas[m] || t[3] < 2 || t[4] < 2 || t[1] < 0 || t[2] < 0 || t[1] + t[3] > d->atlas_w[m] || t[2] + t[4] > d->atlas_h[m]) {
                delete s; return fail(APT_E_INVALID, "apt_scene_create: texture rectangle outside its atlas (or smaller than 2 x 2)");
            }
        }
        std::vector<float> uv(d->uvs, d->uvs + (size_t)N * 6), tf(d->tex_f, d->tex_f + (size_t)O * 6);
        std::vector<int> ti(d->tex_i, d->tex_i + (size_t)O * 15);
        UP(uvs, uv); UP(tex_i, ti); UP(tex_f, tf);
        ds.uvs = s->uvs.as<float>(); ds.tex_i = s->tex_i.as<int>(); ds.tex_f = s->tex_f.as<float>();
        for (int m = 0; m < 3; m++) if (d->atlas[m]) {
            std::vector<float> img(d->atlas[m], d->atlas[m] + (size_t)d->atlas_w[m] * (size_t)d->atlas_h[m] * 3);
            UP(atlas[m], img);
            ds.atlas[m] = s->atlas[m].as<float>(); ds.atlas_w[m] = d->atlas_w[m];
        }
    }
    {   // per-primitive shading records
        std::vector<float> ps((size_t)N * 8, 0.f);
        apt::parallel_for(N, host_thr, [&](int k) {
            float* r = ps.data() + 8 * (size_t)k;
            const int o = prim_obj[(size_t)k];
            int32_t code = sphere[(size_t)k] ? ~o : o, light = d->emitter_id[o];
            const float* src3 = sphere[(size_t)k] ? prec.data() + 9 * (size_t)k : d->normals + 3 * (size_t)k;      // centre | n_g
            r[0] = src3[0]; r[1] = src3[1]; r[2] = src3[2];
            memcpy(&r[3], &code, 4); memcpy(&r[4], &light, 4);
            r[5] = d->bxdf_f[13 * o]; r[6] = d->bxdf_f[13 * o + 1]; r[7] = d->bxdf_f[13 * o + 2];
        });
        UP(prim_shade, ps);
        ds.prim_shade = s->prim_shade.as<float4>();
    }
    {   // participating media: transparent everywhere unless the description carries the tables
        std::vector<DevMedium> md((size_t)O + 1);
        for (int o = 0; o <= O; o++) {
            DevMedium& m = md[(size_t)o]; memset(&m, 0, sizeof(m));
            if (d->med_i && d->med_f) {
                const float* f = d->med_f + 16 * (size_t)o;
                m.type = d->med_i[o]; m.ior = f[0];
                m.u_s = mk3(f[1], f[2], f[3]); m.u_a = mk3(f[4], f[5], f[6]); m.u_e = mk3(f[7], f[8], f[9]);
                m.par = mk3(f[10], f[11], f[12]); m.pdf = mk3(f[13], f[14], f[15]);
                if (m.type < -1 || m.type > 3) { delete s; return fail(APT_E_INVALID, "apt_scene_create: unknown medium type"); }
            } else { m.type = -1; m.ior = (o < O) ? bx[(size_t)o].ior : d->world_ior; m.pdf = mk3(1.f, 0.f, 0.f); }
        }
        UP(med, md);
        ds.med = s->med.as<DevMedium>();
        s->world_scattering = md[(size_t)O].type >= 0;
        for (int o = 0; o < O; o++) {
            if (bx[(size_t)o].is_bsdf && bx[(size_t)o].type < 0) s->has_null_surface = true;
            if (d->obj_aabb) for (int a = 0; a < 3; a++) {
                s->box_min[a] = std::min(s->box_min[a], d->obj_aabb[6 * o + a]); s->box_max[a] = std::max(s->box_max[a], d->obj_aabb[6 * o + 3 + a]);
            }
        }
    }
    memset(&ds.vol, 0, sizeof(ds.vol));
    if (d->vol_i && d->vol_f && d->vol_grid && d->vol_i[0] != 0) {
        const int32_t* vi = d->vol_i; const float* f = d->vol_f;
        if (vi[0] != 2) { delete s; return fail(APT_E_INVALID, "apt_scene_create: only RGB grid volumes (type 2) exist upstream"); }
        if (vi[1] <= 0 || vi[2] <= 0 || vi[3] <= 0 || vi[4] < -1 || vi[4] > 3) { delete s; return fail(APT_E_INVALID, "apt_scene_create: bad grid volume shape or phase type"); }
        if (!(f[21] > 0.f && f[22] > 0.f && f[23] > 0.f)) { delete s; return fail(APT_E_INVALID, "apt_scene_create: grid volume majorants must be positive"); }
        std::vector<float> grid(d->vol_grid, d->vol_grid + (size_t)vi[1] * (size_t)vi[2] * (size_t)vi[3] * 3);
        UP(vol_grid, grid);
        DevVolume& vo = ds.vol;
        vo.type = vi[0]; vo.xres = vi[1]; vo.yres = vi[2]; vo.zres = vi[3];
        vo.albedo = mk3(f[0], f[1], f[2]);
        vo.inv_r0 = mk3(f[3], f[4], f[5]); vo.inv_r1 = mk3(f[6], f[7], f[8]); vo.inv_r2 = mk3(f[9], f[10], f[11]);
        vo.trans = mk3(f[12], f[13], f[14]); vo.mini = mk3(f[15], f[16], f[17]); vo.maxi = mk3(f[18], f[19], f[20]);
        vo.majorant = mk3(f[21], f[22], f[23]); vo.pdf = mk3(f[24], f[25], f[26]);
        vo.ph.type = vi[4]; vo.ph.par = mk3(f[27], f[28], f[29]); vo.ph.pdf = mk3(f[30], f[31], f[32]);
        vo.grid = s->vol_grid.as<float>();
        s->has_volume = true;
    }
#undef UP
    tick("tables + uploads");
    *out = s;
    return APT_OK;
}
APT_EXPORT void apt_scene_destroy(apt_scene* s) { if (s) { (void)hipSetDevice(s->device); delete s; } }

// ---- renderer
static int owned_columns(const apt_render_cfg& c) {
    int n = 0;
    for (int x = 0; x < c.width; x++) if ((x / c.band_width) % c.world_size == c.rank) n++;
    return n;
}

APT_EXPORT void apt_renderer_destroy(apt_renderer* r);
APT_EXPORT int apt_renderer_create(const apt_scene* sc, const apt_render_cfg* cfg, apt_renderer** out) {
    if (!sc || !cfg || !out) return fail(APT_E_INVALID, "apt_renderer_create: null argument");
    apt_render_cfg c = *cfg;
    if (c.width <= 0 || c.height <= 0 || c.max_bounce < 0 || c.num_shadow_ray < 0) return fail(APT_E_INVALID, "apt_renderer_create: bad film / bounce settings");
    // a path's position in its random stream travels in 16 bits of the meta word (23 in the volumetric tracer): bound the draws a path can make
    if (!c.volumetric && (int64_t)(5 * (int64_t)c.num_shadow_ray + 8) * (int64_t)c.max_bounce + 4 >= 65536)
        return fail(APT_E_INVALID, "apt_renderer_create: num_shadow_ray x max_bounce too large for the 16-bit draw index of a path");
    if (c.world_size <= 0) { c.world_size = 1; c.rank = 0; }
    if (c.band_width <= 0) c.band_width = c.width;
    if (c.rank < 0 || c.rank >= c.world_size) return fail(APT_E_INVALID, "apt_renderer_create: rank outside world_size");
    if (c.device != sc->device) return fail(APT_E_INVALID, "apt_renderer_create: renderer and scene must live on the same device");
    HIP_TRY(hipSetDevice(c.device));
    apt_renderer* r = new apt_renderer();
    struct Owner { apt_renderer* r; ~Owner() { if (r) apt_renderer_destroy(r); } } owner{r};     // every early return below releases streams, events and queue pools
    r->scene = sc; r->cfg = c;
    r->n_cols = owned_columns(c);
    // every owned band must be complete except possibly the last: local->global mapping assumes it
    r->npix = r->n_cols * c.height;
    if (r->npix <= 0) { return fail(APT_E_INVALID, "apt_renderer_create: this rank owns no pixels"); }
    int B = c.spp_per_batch;
    r->n_lanes = 3;   // measured on C2: 1 lane 1 827, 2 lanes 2 196, 3 lanes 2 268, 4 lanes 2 178 Msamples/s (round 2, 64 spp per lane-batch); round 5, 32 Mi paths per lane-batch: C2 4 559 / 4 590 / 4 287 with 2 / 3 / 4, C5 1 844 / 2 028 / 2 129 with 1 / 2 / 3; volumetric scenes with null surfaces ran four lanes in rounds 3-4 (the fourth hid the host read-backs of their tails): at this batch size V1 1 374 with three, 1 299 with four
    if (const char* nl = getenv("APT_LANES")) r->n_lanes = std::min(4, std::max(1, atoi(nl)));
    if (B <= 0) {
        // ~32 Mi paths per lane-batch (5-17 GB of queues each, of 288 GB) whatever the tile size: an 8-GPU rank owns 1/8 of the pixels and takes
        // 8x the samples per batch.  Measured 16 Mi -> 32 Mi on one box: C2 4 547 -> 4 685, C3 1 150 -> 1 177, C4 1 840 -> 1 922, C5 1 971 ->
        // 2 078 Msamples/s (half the launches, each twice as long: a launch boundary drains and refills 256 CUs); 48-90 Mi is level or worse.
        // Class-sorted and volumetric renders address (classes x capacity) 16-byte slots with 32-bit byte offsets: the batch stays below that.
        B = (int)((32u << 20) / (uint32_t)r->npix); if (B < 1) B = 1; if (B > 1024) B = 1024;
        const size_t n_cq = c.volumetric ? (size_t)sc->n_classes + 1 : ((sc->n_classes >= 2 && sc->n_classes <= APT_MAX_CLASSES) ? (size_t)sc->n_classes : 1);
        // (the capacity a batch of B really gets, as computed below: whole waves, dealt over the sub-queues)
        int nq_fit = r->nq; if (const char* e = getenv("APT_NQ")) nq_fit = std::min(APT_MAX_NQ, std::max(1, atoi(e)));
        auto cap_of = [&](int b) { const size_t nw = ((size_t)r->npix * (size_t)b + 63) / 64; return ((nw + (size_t)nq_fit - 1) / (size_t)nq_fit) * 64 * (size_t)nq_fit; };
        while (B > 1 && n_cq * cap_of(B) >= ((size_t)1 << 28)) B--;
        // ... and the lanes' queue pools within a third of the memory that is free now (a second renderer beside this one, a device shared by
        // several ranks, a smaller part): ~0.4-0.8 KB per path (upper estimate: SoA queues, packed records, one 64-byte record per class queue)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t per_path = 4u * (64u + 16u * n_cq + 11u * (size_t)std::max(1, c.num_shadow_ray));
            while (B > 1 && (size_t)r->npix * (size_t)B * (size_t)r->n_lanes * per_path > free_b / 3) B = (B + 1) / 2;
        }
    }
    r->spp_batch = B;
    const int S = c.num_shadow_ray;
    if (const char* e = getenv("APT_NQ")) r->nq = std::min(APT_MAX_NQ, std::max(1, atoi(e)));      // tuning knob: sub-queues per queue
    const int nq = r->nq;
    const size_t n_waves = ((size_t)r->npix * (size_t)B + 63) / 64;
    const size_t subcap = ((n_waves + nq - 1) / nq) * 64;            // generate: wave w -> sub-queue w % nq
    const size_t cap = subcap * (size_t)nq;
    const size_t sh_subcap = subcap * (size_t)(S > 0 ? S : 1);
    const size_t sh_cap = sh_subcap * (size_t)nq;
    if (cap >= (1ull << 31)) { return fail(APT_E_INVALID, "apt_renderer_create: batch too large (spp_per_batch * pixels must stay below 2^31)"); }
    for (const ShadeVariant& v : kShadeVariants)
        if ((sc->bx_mask & ~v.bm) == 0 && (sc->src_mask & ~v.sm) == 0) { r->shade = &v; break; }
    if (!r->shade) { return fail(APT_E_INVALID, "apt_renderer_create: scene uses a material or emitter type the kernels do not know"); }
    const bool textured = sc->dev.tex_i != nullptr;
    if (textured) r->shade = &kTexturedShade;                   // the one kernel compiled with texture lookups; unsorted
    Params& p = r->par;
    memcpy(p.cam_r, c.cam_r, sizeof(p.cam_r)); memcpy(p.cam_t, c.cam_t, sizeof(p.cam_t));
    p.inv_focal = c.inv_focal; p.half_w = c.half_w; p.half_h = c.half_h;
    p.W = c.width; p.H = c.height; p.n_cols = r->n_cols; p.npix = r->npix;
    p.band_width = c.band_width; p.rank = c.rank; p.world = c.world_size;
    p.do_crop = c.do_crop; p.sx = c.start_x; p.ex = c.end_x; p.sy = c.start_y; p.ey = c.end_y;
    p.max_bounce = c.max_bounce; p.S = S; p.inv_S = (S > 0) ? 1.f / (float)S : 1.f;
    p.use_rr = c.use_rr; p.use_mis = c.use_mis; p.anti_alias = c.anti_alias; p.stratified = c.stratified; p.two_sides = c.brdf_two_sides;
    p.rr_bounce_th = c.rr_bounce_th; p.rr_threshold = c.rr_threshold; p.seed = c.seed; p.cap = (uint32_t)cap; p.subcap = (uint32_t)subcap; p.nq = nq;
    p.inv_ns = 1.f / (float)sc->n_sources; p.inv_ns1 = (sc->n_sources > 1) ? 1.f / (float)(sc->n_sources - 1) : 1.f;
    // traversal mode: small scenes sweep all primitives wave-uniformly in the reference's brute-force order
    // the tiled sweep pays off once some object has enough primitives that skipping it per ray matters;
    // scenes of spheres and quads only are as fast in the plain wave sweep
    const bool tile_ok = sc->has_aabb && sc->has_sweep && sc->n_objects <= APT_TILE_MAX_OBJECTS && sc->n_prims < 65536 &&
                         APT_TILE_LDS_BYTES(APT_TILE_NT, sc->n_objects) <= 160 * 1024;      // the per-object lists of a tile must fit the CU's LDS (34 objects at 512 threads)
    r->trace_mode = (sc->n_prims <= APT_SWEEP_MAX_PRIMS && sc->has_aabb && sc->has_sweep) ? ((tile_ok && sc->max_obj_prims >= APT_SWEEP_LIST_MIN) ? 2 : 1) : 0;
    if (sc->has_flat) r->trace_mode = 3;                 // fast build: the flat sweep serves every scene small enough to have its records
    if (const char* force = getenv("APT_TRAVERSAL")) {
        if (!strcmp(force, "bvh")) r->trace_mode = 0;
        else if (!strcmp(force, "flat") && sc->has_flat) r->trace_mode = 3;
        else if (!strcmp(force, "sweep") && sc->has_aabb && sc->has_sweep) r->trace_mode = 1;
        else if (!strcmp(force, "tile") && tile_ok) r->trace_mode = 2;
    }
    // flat sweep with several light samples per vertex: the samples are queued by vertex and the shadow kernel adds a vertex's samples with
    // one read-modify-write (stages.hpp k_shadow_flat); otherwise 2-4 samples per vertex add into one radiance plane each
    p.volumetric_flat = c.volumetric ? 1 : 0; p.fix_par = 0;
    p.nee_vm = (!c.volumetric && r->trace_mode == 3 && S > 1) ? 1 : 0;
    if (const char* f = getenv("APT_NEE_VM")) p.nee_vm = (atoi(f) != 0 && !c.volumetric && r->trace_mode == 3 && S > 1) ? 1 : 0;
    p.l_planes = (!p.nee_vm && S >= 2 && S <= 4) ? S : 1;        // stages.hpp APT_EXCLUSIVE_L: one radiance plane per light sample of a vertex
    p.fused = 0;
    p.pix_bits = 0; while ((1u << p.pix_bits) < (uint32_t)r->npix) p.pix_bits++;
    if (((uint64_t)B << p.pix_bits) > (1ull << 32)) { return fail(APT_E_INVALID, "apt_renderer_create: spp_per_batch x pixels does not fit the 32-bit path id"); }
    {   // local pixel -> RNG key (global pixel index), same mapping as local_to_global in stages.hpp
        std::vector<uint32_t> key((size_t)r->npix);
        for (int lp = 0; lp < r->npix; lp++) {
            const int lc = lp / c.height, j = lp % c.height, lb = lc / c.band_width, w_ = lc % c.band_width;
            const int i = (lb * c.world_size + c.rank) * c.band_width + w_;
            key[(size_t)lp] = (uint32_t)(i * c.height + j);
        }
        hipError_t e_ = upload(r->pix_key, key);
        if (e_ != hipSuccess) { return fail(APT_E_HIP, std::string("upload pix_key: ") + hipGetErrorString(e_)); }
        p.pix_key = r->pix_key.as<uint32_t>();
    }
    r->sorted = (sc->n_classes >= 2 && sc->n_classes <= APT_MAX_CLASSES) ? 1 : 0;        // all nine surface models in one scene: the all-models kernel
    if (const char* force = getenv("APT_SORTED")) r->sorted = (atoi(force) != 0 && sc->n_classes >= 1 && sc->n_classes <= APT_MAX_CLASSES) ? 1 : 0;
    if (textured) r->sorted = 0;
    r->volumetric = c.volumetric ? 1 : 0;
    // rays traced by the shade kernel itself (shade_stage.hpp k_shade_traced, "rays traced in place": no extend, shadow or fix-up launch per bounce):
    // flat sweep, one light sample per vertex, one shade kernel; APT_FUSED=0 stages them
    // - unsorted or sorted by material class, any number of light samples per vertex
    {
        const bool can1 = r->trace_mode == 3 && !r->sorted && !c.volumetric && S == 1;
        bool can2 = can1 && r->shade->traced != nullptr;
        p.fused = can2 ? 2 : 0;
        if (const char* f = getenv("APT_FUSED")) {
            const int want = atoi(f);
            if (want == 1) fprintf(stderr, "adapt_mi: APT_FUSED=1 (light samples only traced in place) was retired in round 5; running the staged pipeline (APT_FUSED=0). Use 2 for rays traced in place.\n");
            p.fused = (want >= 2 && can2) ? 2 : 0;
        }
        if (p.fused == 2) { p.nee_vm = 0; p.l_planes = 1; }      // no shadow queue: a vertex's light samples are summed in registers
    }
    if (r->volumetric) {
        if (c.max_bounce > 255) { return fail(APT_E_INVALID, "apt_renderer_create: the volumetric tracer keeps the bounce count in 8 bits (max_bounce <= 255)"); }
        if (!sc->has_aabb) { return fail(APT_E_INVALID, "apt_renderer_create: the volumetric tracer needs the object boxes (world bound)"); }
        r->sorted = 0;
        // shading sorted by EVENT (volumetric.hpp): k_vevent, then one kernel per event queue - the surface classes, the medium last
        r->vevent = 1;
        r->vev_single = (textured || sc->n_classes + 1 > APT_MAX_CLASSES) ? 1 : 0;
        const int n_surf = r->vev_single ? 1 : sc->n_classes;
        r->v_ncls = n_surf + 1;
        const int smi = ((sc->src_mask & ~0x03) == 0) ? 0 : 1, vi = sc->has_volume ? 1 : 0;
        r->vev_all = kVEventAll[smi][vi];
        for (int c = 0; c <= n_surf; c++) r->vev_live[c] = false;
        for (size_t o = 0; o < sc->obj_class.size(); o++)           // which surface queues can receive a hit at all
            if (!sc->obj_null[o]) r->vev_live[r->vev_single ? 0 : sc->obj_class[o]] = true;
        r->vev_live[n_surf] = true;
        for (int g = 0; g < APT_N_VGROUPS; g++) { r->vgroup_fn[g] = kVGroup[smi][vi][g]; for (int k = 0; k < 4; k++) r->vgroup_cls[g][k] = -1; }
        if (!r->vev_single) for (int c = 0; c < n_surf; c++) if (r->vev_live[c]) r->vgroup_cls[kVClassGroup[sc->class_def[c]]][kVClassSlot[sc->class_def[c]]] = c;
        r->vgroup_cls[1][0] = n_surf;                         // the medium's queue is the last one
        for (int a = 0; a < 3; a++) {                         // path_tracer.py:136-138
            p.w_min[a] = std::min(c.cam_t[a], sc->box_min[a]) - 0.1f; p.w_max[a] = std::max(c.cam_t[a], sc->box_max[a]) + 0.1f;
        }
    }
    const int ncls = r->volumetric ? r->v_ncls : (r->sorted ? sc->n_classes : 0);
    r->shade_name = r->shade->name;
    if (p.fused == 2) r->shade_name += " [rays traced in place]";
    if (r->vevent) {
        r->shade_name = std::string(sc->has_volume ? "volumetric + grid volume, sorted by event: medium | " : "volumetric, sorted by event: medium | ");
        if (r->vev_single) r->shade_name += "all surface models";
        else for (int c = 0; c < sc->n_classes; c++) r->shade_name += std::string(c ? "+" : "") + kClassName[sc->class_def[c]];
    } else if (r->sorted) {
        const int smi = ((sc->src_mask & ~0x03) == 0) ? 0 : (((sc->src_mask & ~0x05) == 0) ? 2 : 1);
        r->shade_name = "sorted:";
        for (int g = 0; g < APT_N_GROUPS; g++) { r->group_fn_[g] = kGroupShade[smi][g]; for (int k = 0; k < APT_GROUP_SLOTS; k++) r->group_cls[g][k] = -1; }
        r->shade_name = "sorted, launched in register-footprint groups:";
        for (int c = 0; c < ncls; c++) {
            r->group_cls[kClassGroup[sc->class_def[c]]][kClassSlot[sc->class_def[c]]] = c;
            r->shade_name += std::string(c ? "+" : "") + kClassName[sc->class_def[c]];
        }
    }
    // the kernels address queue slots with 32-bit byte offsets (stages.hpp "Queue addressing")
    if (cap >= ((size_t)1 << 30) || sh_cap >= ((size_t)1 << 30)) { return fail(APT_E_INVALID, "apt_renderer_create: batch too large (queue capacity must stay below 2^30 slots)"); }
    // one pool per lane, carved into the SoA arrays (all 4-byte lanes)
    const bool walk_lists = r->volumetric && sc->has_null_surface;      // light samples that cross null surfaces are re-queued by slot
    const size_t l_planes = (size_t)p.l_planes;
    if (p.fused == 2 && cap >= ((size_t)1 << 28)) { return fail(APT_E_INVALID, "apt_renderer_create: batch too large (rays traced in place address 16-byte slots with 32-bit byte offsets: capacity must stay below 2^28)"); }
    const bool tr_uv = p.fused == 2 && (sc->dev.has_vn || sc->dev.tex_i != nullptr);
    // rays traced in place keep their path records in planes of their own (Queues::tr): the staged pipeline's second ray / state buffers are not carved
    if (((r->sorted && !r->volumetric) || r->vevent) && (size_t)ncls * cap >= ((size_t)1 << 28)) { return fail(APT_E_INVALID, "apt_renderer_create: batch too large (packed class queues address 16-byte slots with 32-bit byte offsets: classes x capacity must stay below 2^28)"); }
    const bool stage_top = true;      // the staging queue lives at the top of the sub-queue's own region (Queues::tr_stage_top)
    const size_t tr_q = (p.fused == 2) ? 1 : 0;      // queues per plane (rays traced in place: unsorted renders, one queue)
    if (p.fused == 2 && tr_q * cap >= ((size_t)1 << 28)) { return fail(APT_E_INVALID, "apt_renderer_create: batch too large (rays traced in place: queues x capacity must stay below 2^28 slots)"); }
    const size_t words = (p.fused == 2 ? (32 + (tr_uv ? 4 : 0)) * cap * tr_q : 0) - (p.fused == 2 ? (6 + 12) * cap : 0) + (r->trace_mode == 3 ? cap + sh_cap : 0) + cap * (6 * 2 + 4 + (3 + 1 + 1 + 1) * 2 + 4 * l_planes) + sh_cap * (3 + 3 + 1 + 3 + 1) + (p.fused == 2 ? 0 : cap * 16 * (size_t)ncls) + (walk_lists ? 2 * sh_cap : 0);
    auto carve = [&](DevBuf& pool, Queues& q) -> hipError_t {
        hipError_t e_ = pool.alloc(words * 4);
        if (e_ != hipSuccess) return e_;
        // zero-filled once: some queue arrays are only written when somebody reads them (the flat extend kernel skips hit_u / hit_v unless
        // vertex normals or textures need the barycentrics), and a kernel that loads them anyway must not see a previous renderer's bytes
        // (hipMemset runs on the null stream and does not wait on the host; the render lanes are non-blocking streams that do not
        // synchronise with it: the hipDeviceSynchronize() after the last pool is carved is what orders the fill before the first launch)
        if ((e_ = hipMemset(pool.p, 0, words * 4)) != hipSuccess) return e_;
        float* w = pool.as<float>();
        auto take = [&](size_t n) { float* x = w; w += n; return x; };
        const bool staged = p.fused != 2;      // (rays traced in place: parity 0 of the ray arrays and the hit arrays still serve apt_intersect)
        for (int k = 0; k < 2; k++) { q.ray_o[k] = (k == 0 || staged) ? take(3 * cap) : nullptr; q.ray_d[k] = (k == 0 || staged) ? take(3 * cap) : nullptr; }
        q.hit_t = take(cap); q.hit_prim = (int*)take(cap); q.hit_u = take(cap); q.hit_v = take(cap);
        for (int k = 0; k < 2; k++) {
            for (int a = 0; a < 4; a++) q.tr[k][a] = (p.fused == 2) ? (float4*)take(4 * cap * tr_q) : nullptr;
            q.tr_uv[k] = tr_uv ? (float2*)take(2 * cap * tr_q) : nullptr;
        }
        q.tr_ncls = (p.fused == 2) ? (int)tr_q - (stage_top ? 0 : 1) : 0; q.tr_stage_top = (p.fused == 2 && stage_top) ? 1 : 0;
        q.fix_ext = (r->trace_mode == 3) ? (uint32_t*)take(cap) : nullptr; q.fix_sh = (r->trace_mode == 3) ? (uint32_t*)take(sh_cap) : nullptr;
        for (int k = 0; k < 2; k++) { q.thr[k] = staged ? take(3 * cap) : nullptr; q.id[k] = staged ? (uint32_t*)take(cap) : nullptr; q.meta[k] = staged ? (uint32_t*)take(cap) : nullptr; q.pdf[k] = staged ? take(cap) : nullptr; }
        q.L = take(4 * cap * l_planes);
        q.sh_o = take(3 * sh_cap); q.sh_d = take(3 * sh_cap); q.sh_tmax = take(sh_cap); q.sh_c = take(3 * sh_cap); q.sh_id = (uint32_t*)take(sh_cap);
        q.sh_cap = (uint32_t)sh_cap; q.sh_subcap = (uint32_t)sh_subcap;
        q.sh_walk[0] = walk_lists ? (uint32_t*)take(sh_cap) : nullptr; q.sh_walk[1] = walk_lists ? (uint32_t*)take(sh_cap) : nullptr;
        q.n_classes = ncls;
        for (int a = 0; a < 4; a++) q.cq[a] = (ncls > 0 && p.fused != 2) ? (float4*)take(4 * cap * (size_t)ncls) : nullptr;      // class queues / the volumetric tracer's event queues: packed planes
        return hipSuccess;
    };
    hipError_t e = carve(r->pool, r->q);
    if (e != hipSuccess) { return fail(APT_E_NOMEM, std::string("queue pool: ") + hipGetErrorString(e)); }
    r->extra.resize((size_t)r->n_lanes - 1);
    for (auto& ln : r->extra) {
        if ((e = carve(ln.pool, ln.q)) != hipSuccess || (e = ln.counters.alloc(sizeof(Counters))) != hipSuccess) { return fail(APT_E_NOMEM, std::string("queue pool (lane): ") + hipGetErrorString(e)); }
    }
    HIP_TRY(hipDeviceSynchronize());
    if ((e = r->counters.alloc(sizeof(Counters))) != hipSuccess || (e = r->accum.alloc((size_t)r->npix * 12)) != hipSuccess ||
        (e = r->scratch.alloc((size_t)r->npix * 12)) != hipSuccess) { return fail(APT_E_NOMEM, std::string("framebuffer: ") + hipGetErrorString(e)); }
    HIP_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    HIP_TRY(hipMemsetAsync(r->counters.p, 0, sizeof(Counters), r->stream));
    HIP_TRY(hipEventCreateWithFlags(&r->fin0, hipEventDisableTiming));
    for (auto& ln : r->extra) {
        HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ln.fin, hipEventDisableTiming));
        HIP_TRY(hipMemsetAsync(ln.counters.p, 0, sizeof(Counters), ln.stream));
        HIP_TRY(hipStreamSynchronize(ln.stream));
    }
    HIP_TRY(hipMemsetAsync(r->accum.p, 0, (size_t)r->npix * 12, r->stream));
    HIP_TRY(hipEventCreate(&r->ev_r0)); HIP_TRY(hipEventCreate(&r->ev_r1));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c.device));
    int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // LDS plan of the BVH walk: the per-lane stack of 8-byte groups.  A node visit leaves at most one group behind (the rest of its
    // hit children), so the stack never holds more groups than the tree has levels.
    {
        LdsPlan pl;
        const int full_depth = sc->wide.max_depth + 2;
        int lds_levels = 7;                                  // 14 KiB per 256-thread workgroup (measured: 6, 10 and 14 levels run alike - the 8-wide tree is at most 8-9 levels deep; 8 through round 5: the eighth level's 2 KiB hold the walk's permutation table now, seven workgroups per CU as before)
        if (const char* sd = getenv("APT_BVH_LDS_LEVELS")) lds_levels = std::max(2, atoi(sd));
        pl.stack_depth = std::min(full_depth, lds_levels);   // deeper levels spill to per-lane global columns (traverse.hpp TravStack)
        r->ovf_levels = full_depth - pl.stack_depth;
        pl.ovf = nullptr; pl.ovf_stride = 0;
        const size_t stack_b = (size_t)pl.stack_depth * BLOCK * 8;
        r->plan = pl;
        r->lds_bytes = stack_b + (size_t)6 * BLOCK * 4 + 2048;      // (+ k_extend_dyn's parked path state: six floats per thread, + the walk kernels' 8 x 256-byte priority-permutation table: stages.hpp make_walk_stack)
        if (r->lds_bytes > 160 * 1024) { return fail(APT_E_INVALID, "apt_renderer_create: BVH too deep for the LDS traversal stack"); }
        int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / r->lds_bytes));
        r->grid_trace = cus * per_cu;
        if (r->trace_mode == 3) {
            // no LDS; the grid is a whole number of rounds of what the register allocation lets a CU hold
            r->lds_bytes = 0; r->trace_items = 2 * BLOCK;
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kShadow[3], BLOCK, 0) != hipSuccess || occ < 1) occ = 4;
            r->grid_trace = cus * std::min(occ, 8) * 2;
        }
        if (r->trace_mode == 1) {
            // persistent grid = a whole number of rounds of what the CU can hold: the sweep kernels allocate 85-89 VGPRs, five 256-thread
            // workgroups per CU, and a grid of 8 per CU ran as one full round plus a 3/5 one (C3, ms per 64 spp, extend / shadow at
            // 4 / 5 / 6 / 8 / 10 / 16 workgroups per CU: 6.4 / 5.7 / 7.2 / 6.2 / 5.6 / 5.8 and 14.0 / 13.6 / 15.0 / 13.8 / 13.3 / 13.3)
            r->lds_bytes = 0;
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kShadow[1], BLOCK, 0) != hipSuccess || occ < 1) occ = 4;
            r->grid_trace = cus * std::min(occ, 8) * 2;
        }
        if (r->trace_mode == 2) {
            r->trace_nt = APT_TILE_NT; r->trace_items = APT_TILE_NT;
            r->lds_bytes = APT_TILE_LDS_BYTES(APT_TILE_NT, sc->n_objects);
            r->lds_bytes_any = APT_TILE_LDS_BYTES_ANY(APT_TILE_NT, sc->n_objects);
            r->grid_trace = cus * (int)std::max<size_t>(1, std::min<size_t>(2048 / APT_TILE_NT, (160 * 1024) / r->lds_bytes));
            // 69 VGPRs allow 7 waves per SIMD = three 8-wave workgroups per CU; the slimmer LDS footprint leaves room for other lanes' kernels
            r->grid_shadow = cus * (int)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / r->lds_bytes_any));
            if (r->lds_bytes > 64 * 1024) {
                HIP_TRY(hipFuncSetAttribute((const void*)kExtend[2][0], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
                HIP_TRY(hipFuncSetAttribute((const void*)kExtend[2][1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
                HIP_TRY(hipFuncSetAttribute((const void*)kShadow[2], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
                HIP_TRY(hipFuncSetAttribute((const void*)kOccluded[2], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
                HIP_TRY(hipFuncSetAttribute((const void*)kVShadow[2], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
            }
        } else if (r->lds_bytes > 64 * 1024) {
            HIP_TRY(hipFuncSetAttribute((const void*)kExtend[0][0], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
            HIP_TRY(hipFuncSetAttribute((const void*)kExtend[0][1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
            HIP_TRY(hipFuncSetAttribute((const void*)kShadow[0], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
            HIP_TRY(hipFuncSetAttribute((const void*)kOccluded[0], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
            HIP_TRY(hipFuncSetAttribute((const void*)kVShadow[0], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
        }
    }
    r->dyn_fetch = (r->trace_mode == 0) ? 1 : 0;         // BVH scenes: closest-hit walk with dynamic ray fetch
    if (const char* f = getenv("APT_DYN_FETCH")) r->dyn_fetch = (atoi(f) != 0 && r->trace_mode == 0) ? 1 : 0;
    if (r->dyn_fetch && r->lds_bytes > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void*)kExtendDyn[0], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
        HIP_TRY(hipFuncSetAttribute((const void*)kExtendDyn[1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
        HIP_TRY(hipFuncSetAttribute((const void*)k_shadow_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
    }
    // streaming stages: persistent grid in 256-thread workgroups per CU.  (rays traced in place: the kernel holds FIVE workgroups per CU since round 5 - one lane 3 440 -> 3 630 Msamples/s on C2 with 5 per CU, three lanes level)  streaming stages, persistent grid in 256-thread workgroups per CU.  Measured (tools/grid_sweep.sh, 2 / 3 / 4 / 5 / 8 per CU): C2 3 385 / 3 350 / 3 378 / 3 312 / 3 279 Msamples/s (a shade kernel holds 4 workgroups per CU; a second round of workgroups only adds a tail), C1 / C3 / C4 / C5 within 1 %, V1 823 / 838 / 842 / 840 / 858
    // (round 6: class-sorted renders too - their lean group holds five workgroups per CU now: C5 2 243 -> 2 261, C3 one lane 1 035 -> 1 103, three lanes level)
    r->grid_small = cus * (r->volumetric ? 8 : ((p.fused == 2 || r->sorted) ? 5 : 4));
    if (const char* g = getenv("APT_GRID_SMALL")) r->grid_small = cus * std::max(1, atoi(g));       // tuning knobs: workgroups per CU
    if (const char* g = getenv("APT_GRID_TRACE")) r->grid_trace = cus * std::max(1, atoi(g));
    if (r->trace_mode != 2) { r->lds_bytes_any = r->lds_bytes; r->grid_shadow = r->grid_trace; }
    if (const char* g = getenv("APT_GRID_SHADOW")) r->grid_shadow = cus * std::max(1, atoi(g));
    r->grid_vshadow = r->grid_trace; r->vshadow_nt = r->trace_nt; r->vshadow_lds = r->lds_bytes;
    r->vshadow_mode = r->trace_mode;
    if (r->trace_mode == 3) {                            // measured: V2 530 -> 500, V3 543 -> 499 Msamples/s with the walk on the flat sweep's one-ray adapter (k_vshadow<3>: half of every packed instruction idle)
        r->vshadow_mode = (tile_ok && sc->max_obj_prims >= APT_SWEEP_LIST_MIN) ? 2 : 1;
        // ... so the flat walk is a kernel of its own: two samples per lane, all segments of a sample in one launch (volumetric.hpp k_vshadow_flat); APT_VSHADOW_FLAT=0: the tiled / wave sweep, for A/B
        const char* vf = getenv("APT_VSHADOW_FLAT");
        if (!(vf && atoi(vf) == 0)) r->vshadow_mode = 4;
        if (r->vshadow_mode == 1) { int occ = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kVShadow[1], BLOCK, 0) != hipSuccess || occ < 1) occ = 4; r->grid_vshadow = cus * std::min(occ, 8) * 2; r->vshadow_lds = 0; }
    }
    if (r->vshadow_mode == 2) {
        r->vshadow_nt = APT_VSHADOW_NT;
        r->vshadow_lds = APT_TILE_LDS_BYTES(APT_VSHADOW_NT, sc->n_objects);
        r->grid_vshadow = cus * std::max(1, std::min((int)((160 * 1024) / r->vshadow_lds), (APT_VSHADOW_WAVES * 4) / (APT_VSHADOW_NT / 64)));
        if (r->vshadow_lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kVShadow[2], hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->vshadow_lds));
    }
    if (r->vshadow_mode == 4) {                          // (the flat trace kernels' grid and tile: r->grid_trace was sized for trace_mode 3 above)
        r->vshadow_nt = BLOCK; r->vshadow_lds = 0; r->grid_vshadow = r->grid_trace;
    }
    if (const char* g = getenv("APT_GRID_VSHADOW")) r->grid_vshadow = cus * std::max(1, atoi(g));
    r->grid_vshadow = ((r->grid_vshadow + nq - 1) / nq) * nq;
    r->grid_fix = nq * 8;
    if (getenv("APT_DEBUG_GRID")) fprintf(stderr, "[grid] trace mode %d grid %d nt %d lds %zu | shadow grid %d lds %zu | vshadow mode %d grid %d nt %d lds %zu | small %d\n", r->trace_mode, r->grid_trace, r->trace_nt, r->lds_bytes, r->grid_shadow, r->lds_bytes_any, r->vshadow_mode, r->grid_vshadow, r->vshadow_nt, r->vshadow_lds, r->grid_small);
    r->grid_trace = ((r->grid_trace + nq - 1) / nq) * nq;          // persistent grids are multiples of nq
    r->grid_shadow = ((r->grid_shadow + nq - 1) / nq) * nq;
    r->grid_small = ((r->grid_small + nq - 1) / nq) * nq;
    for (auto& ln : r->extra) ln.plan = r->plan;
    if (r->trace_mode == 0 && r->ovf_levels > 0) {
        const int max_grid = std::max(r->grid_trace, std::max(r->grid_shadow, r->grid_vshadow));
        const size_t stride = (size_t)max_grid * BLOCK, bytes = stride * (size_t)r->ovf_levels * 8;
        if ((e = r->ovf.alloc(bytes)) != hipSuccess) { return fail(APT_E_NOMEM, std::string("traversal stack spill: ") + hipGetErrorString(e)); }
        r->plan.ovf = r->ovf.as<uint2>(); r->plan.ovf_stride = (int)stride;
        for (auto& ln : r->extra) {
            if ((e = ln.ovf.alloc(bytes)) != hipSuccess) { return fail(APT_E_NOMEM, std::string("traversal stack spill (lane): ") + hipGetErrorString(e)); }
            ln.plan = r->plan; ln.plan.ovf = ln.ovf.as<uint2>();
        }
    }
    HIP_TRY(hipStreamSynchronize(r->stream));
    owner.r = nullptr;
    *out = r;
    return APT_OK;
}
APT_EXPORT void apt_renderer_destroy(apt_renderer* r) {
    if (!r) return;
    (void)hipSetDevice(r->cfg.device);
    for (auto& ln : r->extra) { if (ln.stream) { (void)hipStreamSynchronize(ln.stream); (void)hipStreamDestroy(ln.stream); } if (ln.fin) (void)hipEventDestroy(ln.fin); }
    if (r->stream) { (void)hipStreamSynchronize(r->stream); (void)hipStreamDestroy(r->stream); }
    if (r->fin0) (void)hipEventDestroy(r->fin0);
    for (auto& ev : r->pending) { if (ev.own_a) (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto& ev : r->free_events) (void)hipEventDestroy(ev);
    if (r->ev_r0) (void)hipEventDestroy(r->ev_r0);
    if (r->ev_r1) (void)hipEventDestroy(r->ev_r1);
    delete r;
}

// persistent grid: enough workgroups for n items, capped, and a multiple of the sub-queue count
static int grid_for(size_t n, int cap_blocks, int nq, int nt = BLOCK) {
    size_t b = (n + nt - 1) / nt;
    b = ((b + nq - 1) / nq) * nq;
    if (b < (size_t)nq) b = nq;
    return (int)(b < (size_t)cap_blocks ? b : (size_t)cap_blocks);
}

// Per-launch timing (cfg.profile): a launch is bracketed by two events on its stream.  Launches that follow each other directly on one
// stream share the event between them - one hipEventRecord per launch instead of two (each costs the stream a few microseconds: 52 of them
// per 7 ms lane-batch were 4 % of C2's rate) - so a duration then includes the gap to the previous kernel's end, i.e. errs on the long side.
// Anything else enqueued on the stream (a memset, a wait on another lane's event) must call unchain() first, or its time is charged to the
// next kernel.
static void unchain(apt_renderer* r, hipStream_t st) {
    for (auto& c : r->chain) if (c.first == st) c.second = nullptr;
}
static hipEvent_t take_event(apt_renderer* r) {
    hipEvent_t e = nullptr;
    if (!r->free_events.empty()) { e = r->free_events.back(); r->free_events.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
}
struct LaunchTimer {      // brackets one kernel launch with events when profiling is on
    apt_renderer* r; int kernel; EventPair ev{}; bool on; hipStream_t st;
    LaunchTimer(apt_renderer* r_, int k, hipStream_t stream = nullptr, bool count = true) : r(r_), kernel(k), on(r_->cfg.profile != 0), st(stream ? stream : r_->stream) {
        if (count) r->launches[k]++;                         // (a fix-up launch adds its time to its stage's bucket, not a launch to its count)
        if (!on) return;
        ev.kernel = k; ev.a = nullptr; ev.own_a = false;
        for (auto& c : r->chain) if (c.first == st) ev.a = c.second;
        if (!ev.a) { ev.a = take_event(r); ev.own_a = true; (void)hipEventRecord(ev.a, st); }
        ev.b = take_event(r);
    }
    ~LaunchTimer() {
        if (!on) return;
        (void)hipEventRecord(ev.b, st); r->pending.push_back(ev);
        bool found = false;
        for (auto& c : r->chain) if (c.first == st) { c.second = ev.b; found = true; }
        if (!found) r->chain.push_back({st, ev.b});
    }
};

static int resolve_events(apt_renderer* r) {
    HIP_TRY(hipStreamSynchronize(r->stream));
    for (auto& ev : r->pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        r->kernel_ms[ev.kernel] += ms;
    }
    for (auto& ev : r->pending) { if (ev.own_a) r->free_events.push_back(ev.a); r->free_events.push_back(ev.b); }
    r->pending.clear();
    r->chain.clear();
    if (r->render_pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r->ev_r0, r->ev_r1));
        r->render_ms += ms; r->render_pending = false;
    }
    return APT_OK;
}

// Samples per lane-batch of one render call: the call's samples are split into EQUAL batches of at most the renderer's batch size, a whole
// number of them per render lane (1024 spp at 128 per batch and three lanes: nine batches of 114, not eight of 128 - the lane that would
// have run two of the eight idles a third of the step: C3 1 137 -> 1 177 Msamples/s).  The image does not depend on the split: k_finalize
// adds a pixel's samples to the framebuffer one by one, in sample order, whatever batch they came in.
static int lane_batch(const apt_renderer* r, int32_t n_spp) {
    if (n_spp <= 0) return 1;                       // nothing to split (apt_render(r, 0) is a no-op, not a division by zero)
    int n_batches = (n_spp + r->spp_batch - 1) / r->spp_batch;
    n_batches = ((n_batches + r->n_lanes - 1) / r->n_lanes) * r->n_lanes;
    return std::max(1, std::min(r->spp_batch, (n_spp + n_batches - 1) / n_batches));
}

// Volumetric render (VolumeRenderer.render x n_spp).  Batches go round-robin over the render lanes like the surface tracer's, in
// rounds: first every lane of the round gets its generate + max_bounce iterations (asynchronous), then, lane by lane, the
// null-surface tail (live count read back, two more iterations while paths remain) and the ordered finalize.  While the host
// waits on one lane's tail the other lanes are still busy with their main iterations.
static int render_volumetric(apt_renderer* r, int32_t n_spp) {
    const DevScene& sc = r->scene->dev;
    const int nq = r->nq;
    int done = 0;
    hipEvent_t prev_fin = nullptr;
    const int batch_cap = lane_batch(r, n_spp);
    struct Issued { int li, cur; Params p; size_t total; };
    while (done < n_spp) {
        std::vector<Issued> round;
        for (int li = 0; li < r->n_lanes && done < n_spp; li++) {
            const int B = (n_spp - done < batch_cap) ? (n_spp - done) : batch_cap;
            hipStream_t st = li ? r->extra[(size_t)li - 1].stream : r->stream;
            const Queues& q = li ? r->extra[(size_t)li - 1].q : r->q;
            Counters* cnt = li ? r->extra[(size_t)li - 1].counters.as<Counters>() : r->counters.as<Counters>();
            Issued is; is.li = li; is.cur = 0; is.p = r->par; is.p.cnt_base = r->cnt; is.p.spp_batch = B; is.total = (size_t)r->npix * (size_t)B;
            unchain(r, st); HIP_TRY(hipMemsetAsync(cnt, 0, offsetof(Counters, stats), st));
            { LaunchTimer t(r, 0, st); hipLaunchKernelGGL(k_generate, dim3(grid_for(is.total, r->grid_small, 1)), dim3(BLOCK), 0, st, is.p, q, cnt); }
            round.push_back(is);
            r->cnt += B; done += B;
        }
        auto iterate = [&](Issued& is, int n_iter) -> int {
            hipStream_t st = is.li ? r->extra[(size_t)is.li - 1].stream : r->stream;
            const Queues& q = is.li ? r->extra[(size_t)is.li - 1].q : r->q;
            Counters* cnt = is.li ? r->extra[(size_t)is.li - 1].counters.as<Counters>() : r->counters.as<Counters>();
            const LdsPlan& lane_plan = is.li ? r->extra[(size_t)is.li - 1].plan : r->plan;
            for (int b = 0; b < n_iter; b++) {
                if (r->dyn_fetch) { unchain(r, st); HIP_TRY(hipMemsetAsync(cnt->n_work[0], 0, sizeof(cnt->n_work[0]), st)); }
                { LaunchTimer t(r, 1, st); hipLaunchKernelGGL(r->dyn_fetch ? kExtendDyn[r->sorted] : (r->trace_mode == 3 ? kExtendFlatHot[r->sorted] : kExtend[r->trace_mode][r->sorted]), dim3(grid_for(is.total, r->grid_trace, nq, r->trace_items)), dim3(r->trace_nt), r->lds_bytes, st, sc, is.p, q, cnt, is.cur, (const uint32_t*)cnt->n_active[is.cur], lane_plan); }
                if (r->trace_mode == 3) { LaunchTimer t(r, 1, st, false); hipLaunchKernelGGL(kFixFlat[r->sorted], dim3(r->grid_fix), dim3(BLOCK), 0, st, sc, is.p, q, cnt, is.cur, (const uint32_t*)cnt->n_active[is.cur], lane_plan); }
                {
                    // what every path does this iteration (roulette, hit or world box, free path), then one kernel per event queue
                    { LaunchTimer t(r, 2, st); hipLaunchKernelGGL(kVEvent[r->scene->has_volume ? 1 : 0], dim3(grid_for(is.total, r->grid_small, nq)), dim3(BLOCK), 0, st, sc, is.p, q, cnt, is.cur, r->vev_single); }
                    if (r->vev_single && r->vev_live[0]) { LaunchTimer t(r, 2, st); hipLaunchKernelGGL(r->vev_all, dim3(grid_for(is.total, r->grid_small, nq)), dim3(BLOCK), 0, st, sc, is.p, q, cnt, 0, is.cur); }      // (one surface queue, the all-models kernel)
                    for (int g = 0; g < APT_N_VGROUPS; g++) {      // one launch per register-footprint group of event queues
                        VGroupIn gi; bool any = false;
                        for (int k = 0; k < 4; k++) { gi.cls[k] = r->vgroup_cls[g][k]; any = any || gi.cls[k] >= 0; }
                        if (!any) continue;
                        LaunchTimer t(r, 2, st); hipLaunchKernelGGL(r->vgroup_fn[g], dim3(grid_for(is.total, r->grid_small, nq)), dim3(BLOCK), 0, st, sc, is.p, q, cnt, gi, is.cur);
                    }
                    if (is.p.S <= 0) { unchain(r, st); HIP_TRY(hipMemsetAsync(cnt->n_cls, 0, sizeof(cnt->n_cls), st)); }     // normally k_vshadow's first pass recycles these
                }
                if (is.p.S > 0) {
                    const int n_pass = r->scene->has_null_surface ? (r->vshadow_mode == 4 ? APT_VSHADOW_FLAT_PASSES : 7) : 1;       // track_ray walks at most seven segments (vpt.py:113); the flat walk's last launch walks the later segments itself
                    const int items = r->vshadow_mode == 4 ? 2 * r->vshadow_nt : r->vshadow_nt;        // queue entries per workgroup tile
                    for (int pass = 0; pass < n_pass; pass++) {
                        LaunchTimer t(r, 3, st);
                        hipLaunchKernelGGL(kVShadow[r->vshadow_mode], dim3(grid_for(is.total * (size_t)is.p.S, r->grid_vshadow, nq, items)), dim3(r->vshadow_nt), r->vshadow_lds, st, sc, is.p, q, cnt, lane_plan, pass);
                    }
                }
                is.cur ^= 1;
            }
            return APT_OK;
        };
        // an iteration ends a path or counts a bounce, except a null-surface pass-through: max_bounce iterations finish every
        // path that met no null surface
        for (Issued& is : round) if (int rc = iterate(is, std::max(1, is.p.max_bounce))) return rc;      // the loop body runs once even with max_bounce = 0 (vpt.py:161-245)
        for (Issued& is : round) {
            hipStream_t st = is.li ? r->extra[(size_t)is.li - 1].stream : r->stream;
            const Queues& q = is.li ? r->extra[(size_t)is.li - 1].q : r->q;
            Counters* cnt = is.li ? r->extra[(size_t)is.li - 1].counters.as<Counters>() : r->counters.as<Counters>();
            hipEvent_t fin = is.li ? r->extra[(size_t)is.li - 1].fin : r->fin0;
            if (r->scene->has_null_surface) {
                unsigned long long live = 0;
                for (int k = 0; k < 4096; k++) {
                    unchain(r, st); HIP_TRY(hipMemcpyAsync(r->host_counters.n_active[is.cur], cnt->n_active[is.cur], sizeof(cnt->n_active[is.cur]), hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    live = 0;
                    for (int sq = 0; sq < nq; sq++) live += r->host_counters.n_active[is.cur][sq * CNT_PAD];
                    if (!live) break;
                    if (int rc = iterate(is, 2)) return rc;
                }
                if (live) return fail(APT_E_STATE, "apt_render: paths still crossing null surfaces after 8192 extra iterations (a closed loop of null surfaces?)");
            }
            if (prev_fin && r->n_lanes > 1) { unchain(r, st); HIP_TRY(hipStreamWaitEvent(st, prev_fin, 0)); }
            { LaunchTimer t(r, 4, st); hipLaunchKernelGGL(k_finalize, dim3(grid_for((size_t)r->npix, r->grid_small, 1)), dim3(BLOCK), 0, st, is.p, q, r->accum.as<float>()); }
            HIP_TRY(hipEventRecord(fin, st));
            prev_fin = fin;
            HIP_TRY(hipGetLastError());
        }
    }
    return APT_OK;
}

static int render_impl(apt_renderer* r, int32_t n_spp);
APT_EXPORT int apt_render(apt_renderer* r, int32_t n_spp) {
    if (!r || n_spp < 0) return fail(APT_E_INVALID, "apt_render: bad argument");
    HIP_TRY(hipSetDevice(r->cfg.device));
    if (r->render_pending) { if (int rc = resolve_events(r)) return rc; }
    const int cnt0 = r->cnt;
    const int rc = render_impl(r, n_spp);
    if (rc != APT_OK) {
        // a launch failed half-way: let every lane drain, forget the samples that were not completed, keep the error message
        const std::string msg = g_err;
        for (auto& ln : r->extra) if (ln.stream) (void)hipStreamSynchronize(ln.stream);
        if (r->stream) (void)hipStreamSynchronize(r->stream);
        r->cnt = cnt0; r->render_pending = false;
        g_err = msg;
    }
    return rc;
}
static int render_impl(apt_renderer* r, int32_t n_spp) {
    const DevScene& sc = r->scene->dev;
    HIP_TRY(hipEventRecord(r->ev_r0, r->stream));
    for (auto& ln : r->extra) { unchain(r, ln.stream); HIP_TRY(hipStreamWaitEvent(ln.stream, r->ev_r0, 0)); }     // lanes start after whatever the main stream did before
    if (r->volumetric) {
        if (int rc = render_volumetric(r, n_spp)) return rc;
        for (auto& ln : r->extra) { unchain(r, r->stream); HIP_TRY(hipEventRecord(ln.fin, ln.stream)); HIP_TRY(hipStreamWaitEvent(r->stream, ln.fin, 0)); }
        HIP_TRY(hipEventRecord(r->ev_r1, r->stream));
        r->render_pending = true;
        return APT_OK;
    }
    int done = 0, batch = 0;
    hipEvent_t prev_fin = nullptr;
    const int si = r->sorted ? 1 : 0;            // (sorted: extend appends every hit path's record to the packed queue of its material class, Queues::cq)
    // a call smaller than one full round of lane-batches is split evenly so that every lane has work
    const int batch_cap = lane_batch(r, n_spp);
    while (done < n_spp) {
        const int B = (n_spp - done < batch_cap) ? (n_spp - done) : batch_cap;
        const int li = batch % r->n_lanes;
        hipStream_t st = li ? r->extra[(size_t)li - 1].stream : r->stream;
        const Queues& q = li ? r->extra[(size_t)li - 1].q : r->q;
        Counters* cnt = li ? r->extra[(size_t)li - 1].counters.as<Counters>() : r->counters.as<Counters>();
        hipEvent_t fin = li ? r->extra[(size_t)li - 1].fin : r->fin0;
        const LdsPlan& lane_plan = li ? r->extra[(size_t)li - 1].plan : r->plan;
        Params p = r->par; p.cnt_base = r->cnt; p.spp_batch = B;
        const size_t total = (size_t)r->npix * (size_t)B;
        unchain(r, st); HIP_TRY(hipMemsetAsync(cnt, 0, offsetof(Counters, stats), st));   // queue counters only; statistics keep accumulating
        const int nq = r->nq;
#if APT_FAST
        if (p.fused == 2) {
            // rays traced in place (shade_stage.hpp): generate and every bounce are ONE launch each; the rare rays that need the reference-order code
            // are served by the next launch's prologue, the last bounce's deferred light samples by one fix-up launch at the end
            { LaunchTimer t(r, 0, st); hipLaunchKernelGGL(k_generate_trace, dim3(grid_for(total, r->grid_small, 1)), dim3(BLOCK), 0, st, sc, p, q, cnt); }
            int cur = 0;
            for (int b = 0; b < p.max_bounce; b++) {
                { LaunchTimer t(r, 2, st); hipLaunchKernelGGL(r->shade->traced, dim3(grid_for(total, r->grid_small, nq)), dim3(BLOCK), 0, st, sc, p, q, cnt, cur, b); }      // (the records are Queues::tr[cur]: one queue for the scene)
                cur ^= 1;
            }
            if (p.max_bounce > 0) { LaunchTimer t(r, 3, st, false); hipLaunchKernelGGL(kFixFlat[0], dim3(r->grid_fix), dim3(BLOCK), 0, st, sc, p, q, cnt, cur, (const uint32_t*)cnt->n_active[cur], lane_plan); }
        } else
#endif
        {
        { LaunchTimer t(r, 0, st); hipLaunchKernelGGL(k_generate, dim3(grid_for(total, r->grid_small, 1)), dim3(BLOCK), 0, st, p, q, cnt); }
        int cur = 0;
        for (int b = 0; b < p.max_bounce; b++) {
            // (the walk kernels reset each other's work counters - k_shadow_dyn the next bounce's closest-hit counter, k_extend_dyn this bounce's any-hit counter; without light samples the host does)
            if (r->dyn_fetch && !(p.S > 0)) { unchain(r, st); HIP_TRY(hipMemsetAsync(cnt->n_work[0], 0, sizeof(cnt->n_work[0]), st)); }
            { LaunchTimer t(r, 1, st); hipLaunchKernelGGL(r->dyn_fetch ? kExtendDyn[si] : (r->trace_mode == 3 ? kExtendFlatHot[si] : kExtend[r->trace_mode][si]), dim3(grid_for(total, r->grid_trace, nq, r->trace_items)), dim3(r->trace_nt), r->lds_bytes, st, sc, p, q, cnt, cur, (const uint32_t*)cnt->n_active[cur], lane_plan); }
            if (r->trace_mode == 3) { LaunchTimer t(r, 1, st, false); hipLaunchKernelGGL(kFixFlat[si], dim3(r->grid_fix), dim3(BLOCK), 0, st, sc, p, q, cnt, cur, (const uint32_t*)cnt->n_active[cur], lane_plan); }
            if (!r->sorted) {
                ShadeIn in = {q.ray_o[cur], q.ray_d[cur], q.thr[cur], q.id[cur], q.meta[cur], q.pdf[cur],
                              q.hit_t, q.hit_prim, q.hit_u, q.hit_v, (const uint32_t*)cnt->n_active[cur]};
                LaunchTimer t(r, 2, st); hipLaunchKernelGGL(r->shade->fn, dim3(grid_for(total, r->grid_small, nq)), dim3(BLOCK), 0, st, sc, p, q, cnt, in, cur, b);
            } else {
                for (int g = 0; g < APT_N_GROUPS; g++) {       // one launch per register-footprint group: its workgroups walk the member classes' queues one after the other
                    GroupIn gi; bool any = false;
                    for (int k = 0; k < APT_GROUP_SLOTS; k++) { const int c = r->group_cls[g][k]; gi.cls[k] = c; gi.counts[k] = c >= 0 ? (const uint32_t*)cnt->n_cls[c] : nullptr; any = any || c >= 0; }
                    if (!any) continue;
                    LaunchTimer t(r, 2, st); hipLaunchKernelGGL(r->group_fn_[g], dim3(grid_for(total, r->grid_small, nq)), dim3(BLOCK), 0, st, sc, p, q, cnt, gi, cur, b);
                }
                if (p.S <= 0) { unchain(r, st); HIP_TRY(hipMemsetAsync(cnt->n_cls, 0, sizeof(cnt->n_cls), st)); }      // normally k_shadow recycles these
            }
            if (p.fused) { /* the shade kernel traced the light samples itself */ }
            else if (p.S > 0 && r->dyn_fetch) {
                LaunchTimer t(r, 3, st); hipLaunchKernelGGL(k_shadow_dyn, dim3(grid_for(total * (size_t)p.S, r->grid_shadow, nq, r->trace_items)), dim3(r->trace_nt), r->lds_bytes_any, st, sc, p, q, cnt, lane_plan);
            } else if (p.S > 0) {
                p.fix_par = cur;
                LaunchTimer t(r, 3, st); hipLaunchKernelGGL(kShadow[r->trace_mode], dim3(grid_for(total * (size_t)p.S, r->grid_shadow, nq, r->trace_items)), dim3(r->trace_nt), r->lds_bytes_any, st, sc, p, q, cnt, lane_plan);
            }
            cur ^= 1;
        }
        if (r->trace_mode == 3 && p.S > 0 && p.max_bounce > 0) {     // the last bounce's shadow list (the extend list of this parity is empty)
            LaunchTimer t(r, 3, st, false); hipLaunchKernelGGL(kFixFlat[si], dim3(r->grid_fix), dim3(BLOCK), 0, st, sc, p, q, cnt, cur, (const uint32_t*)cnt->n_active[cur], lane_plan);
        }
        }
        // the framebuffer is shared: batch k's samples are added after batch k-1's, whichever lanes they ran on
        if (prev_fin && r->n_lanes > 1) { unchain(r, st); HIP_TRY(hipStreamWaitEvent(st, prev_fin, 0)); }
        { LaunchTimer t(r, 4, st); hipLaunchKernelGGL(k_finalize, dim3(grid_for((size_t)r->npix, r->grid_small, 1)), dim3(BLOCK), 0, st, p, q, r->accum.as<float>()); }
        HIP_TRY(hipEventRecord(fin, st));
        prev_fin = fin;
        HIP_TRY(hipGetLastError());
        r->cnt += B; done += B; batch++;
    }
    for (auto& ln : r->extra) { unchain(r, r->stream); HIP_TRY(hipEventRecord(ln.fin, ln.stream)); HIP_TRY(hipStreamWaitEvent(r->stream, ln.fin, 0)); }   // join
    HIP_TRY(hipEventRecord(r->ev_r1, r->stream));
    r->render_pending = true;
    return APT_OK;
}
APT_EXPORT int apt_synchronize(apt_renderer* r) {
    if (!r) return fail(APT_E_INVALID, "apt_synchronize: null handle");
    HIP_TRY(hipSetDevice(r->cfg.device));
    return resolve_events(r);
}
APT_EXPORT int apt_tile_shape(const apt_renderer* r, int32_t* n_cols, int32_t* height) {
    if (!r) return fail(APT_E_INVALID, "apt_tile_shape: null handle");
    if (n_cols) *n_cols = r->n_cols;
    if (height) *height = r->cfg.height;
    return APT_OK;
}
APT_EXPORT int apt_read_pixels(apt_renderer* r, float* out) {
    if (!r || !out) return fail(APT_E_INVALID, "apt_read_pixels: bad argument");
    HIP_TRY(hipSetDevice(r->cfg.device));
    const uint32_t n = (uint32_t)r->npix * 3u;
    const float inv = (float)(r->cnt > 0 ? r->cnt : 1);     // before the first sample the image is all zero
    hipLaunchKernelGGL(k_divide, dim3((n + 255) / 256), dim3(256), 0, r->stream, r->accum.as<float>(), r->scratch.as<float>(), n, inv);
    HIP_TRY(hipMemcpyAsync(out, r->scratch.p, (size_t)n * 4, hipMemcpyDeviceToHost, r->stream));
    return resolve_events(r);
}
APT_EXPORT int apt_get_accum(apt_renderer* r, float* out, int32_t* cnt) {
    if (!r || !out) return fail(APT_E_INVALID, "apt_get_accum: bad argument");
    HIP_TRY(hipSetDevice(r->cfg.device));
    HIP_TRY(hipMemcpyAsync(out, r->accum.p, (size_t)r->npix * 12, hipMemcpyDeviceToHost, r->stream));
    if (cnt) *cnt = r->cnt;
    return resolve_events(r);
}
APT_EXPORT int apt_set_accum(apt_renderer* r, const float* in, int32_t cnt) {
    if (!r || !in || cnt < 0) return fail(APT_E_INVALID, "apt_set_accum: bad argument");
    HIP_TRY(hipSetDevice(r->cfg.device));
    HIP_TRY(hipMemcpyAsync(r->accum.p, in, (size_t)r->npix * 12, hipMemcpyHostToDevice, r->stream));
    r->cnt = cnt;
    return resolve_events(r);
}
APT_EXPORT int apt_reset(apt_renderer* r) {
    if (!r) return fail(APT_E_INVALID, "apt_reset: null handle");
    HIP_TRY(hipSetDevice(r->cfg.device));
    HIP_TRY(hipMemsetAsync(r->accum.p, 0, (size_t)r->npix * 12, r->stream));
    HIP_TRY(hipMemsetAsync(r->counters.p, 0, sizeof(Counters), r->stream));
    for (auto& ln : r->extra) HIP_TRY(hipMemsetAsync(ln.counters.p, 0, sizeof(Counters), r->stream));      // lanes are idle between render calls
    r->cnt = 0;
    if (int rc = resolve_events(r)) return rc;
    for (int k = 0; k < APT_N_KERNELS; k++) { r->kernel_ms[k] = 0; r->launches[k] = 0; }
    r->render_ms = 0;
    return APT_OK;
}
APT_EXPORT int apt_get_stats(apt_renderer* r, apt_stats* out) {
    if (!r || !out) return fail(APT_E_INVALID, "apt_get_stats: bad argument");
    HIP_TRY(hipSetDevice(r->cfg.device));
    if (int rc = resolve_events(r)) return rc;
    std::vector<unsigned long long> st((size_t)APT_MAX_NQ * 16), st2((size_t)APT_MAX_NQ * 16);
    HIP_TRY(hipMemcpy(st.data(), (const char*)r->counters.p + offsetof(Counters, stats), st.size() * 8, hipMemcpyDeviceToHost));
    for (auto& ln : r->extra) {
        HIP_TRY(hipMemcpy(st2.data(), (const char*)ln.counters.p + offsetof(Counters, stats), st2.size() * 8, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < st.size(); k++) st[k] += st2[k];
    }
    memset(out, 0, sizeof(*out));
    auto sum = [&](int k) { unsigned long long a = 0; for (int q = 0; q < APT_MAX_NQ; q++) a += st[(size_t)q * 16 + k]; return (int64_t)a; };
    out->n_samples = sum(ST_SAMPLES); out->n_extend = sum(ST_EXTEND); out->n_shade = sum(ST_SHADE);
    out->n_shadow = sum(ST_SHADOW); out->n_shadow_traced = sum(ST_SHADOW_TRACED); out->n_lit = sum(ST_LIT);
    out->n_draws = sum(ST_DRAWS); out->n_poisoned = sum(ST_POISON); out->n_track = sum(ST_TRACK);
#if defined(APT_WALK_STATS) || defined(APT_NEAR_STATS)
    const int64_t n_overflow = 0;                // profiling builds keep their counters in stats[8..15], which overlaps ST_OVERFLOW
#else
    const int64_t n_overflow = sum(ST_OVERFLOW);
#endif
#ifdef APT_NEAR_STATS
    fprintf(stderr, "[near stats] shaded vertices %lld, of which within 2e-3 of the vertex before them %lld\n", (long long)out->n_shade, (long long)sum(14));
#endif
#ifdef APT_WALK_STATS
    {
        unsigned long long wd[2][8], w2[2][8];
        (void)hipMemcpy(wd, (const char*)r->counters.p + offsetof(Counters, wdbg), sizeof(wd), hipMemcpyDeviceToHost);
        for (auto& ln : r->extra) { (void)hipMemcpy(w2, (const char*)ln.counters.p + offsetof(Counters, wdbg), sizeof(w2), hipMemcpyDeviceToHost); for (int a = 0; a < 2; a++) for (int k = 0; k < 8; k++) wd[a][k] += w2[a][k]; }
        for (int a = 0; a < 2; a++) {
            const double it = (double)(wd[a][0] + wd[a][2]);
            fprintf(stderr, "[walk sched] %s: node iterations %llu at %.1f lanes, primitive iterations %llu at %.1f lanes, %.1f lanes hold a ray per iteration; %llu refills claiming %.1f rays each\n", a ? "any-hit" : "closest-hit",
                    wd[a][0], (double)wd[a][1] / std::max(1.0, (double)wd[a][0]), wd[a][2], (double)wd[a][3] / std::max(1.0, (double)wd[a][2]), (double)wd[a][4] / std::max(1.0, it), wd[a][5], (double)wd[a][6] / std::max(1.0, (double)wd[a][5]));
        }
    }
    fprintf(stderr, "[walk stats] closest-hit rays %lld: %.2f node steps, %.2f primitive tests per ray | shadow rays %lld: %.2f node steps, %.2f primitive tests per ray\n",
            (long long)out->n_extend, (double)sum(10) / (double)std::max<int64_t>(1, out->n_extend), (double)sum(11) / (double)std::max<int64_t>(1, out->n_extend),
            (long long)out->n_shadow_traced, (double)sum(12) / (double)std::max<int64_t>(1, out->n_shadow_traced), (double)sum(13) / (double)std::max<int64_t>(1, out->n_shadow_traced));
#endif
    for (int k = 0; k < APT_N_KERNELS; k++) { out->launches[k] = r->launches[k]; out->kernel_ms[k] = r->kernel_ms[k]; }
    out->render_ms = r->render_ms;
    if (n_overflow > 0) return fail(APT_E_STATE, "apt_get_stats: " + std::to_string((long long)n_overflow) + " volumetric path(s) drew more than 2^23 random numbers: the draw index wrapped and those paths re-used part of their stream");
    return APT_OK;
}
APT_EXPORT int apt_device_ptr(apt_renderer* r, void** accum_dev, int32_t* cnt) {
    if (!r || !accum_dev) return fail(APT_E_INVALID, "apt_device_ptr: bad argument");
    *accum_dev = r->accum.p;
    if (cnt) *cnt = r->cnt;
    return APT_OK;
}
APT_EXPORT int apt_stream(apt_renderer* r, void** hip_stream) {
    if (!r || !hip_stream) return fail(APT_E_INVALID, "apt_stream: bad argument");
    *hip_stream = (void*)r->stream;
    return APT_OK;
}

// ---- unit entry points
APT_EXPORT int apt_intersect(apt_renderer* r, int32_t n, const float* o, const float* d, int32_t* prim_out, float* t_out, float* uv_out) {
    if (!r || n <= 0 || !o || !d || !prim_out || !t_out) return fail(APT_E_INVALID, "apt_intersect: bad argument");
    if ((uint32_t)n > r->par.cap) return fail(APT_E_INVALID, "apt_intersect: more rays than the renderer's queue capacity");
    HIP_TRY(hipSetDevice(r->cfg.device));
    if (int rc = resolve_events(r)) return rc;
    const uint32_t cap = r->par.cap;
    std::vector<float> so((size_t)n), sd((size_t)n);
    for (int c = 0; c < 3; c++) {
        for (int k = 0; k < n; k++) { so[(size_t)k] = o[3 * k + c]; sd[(size_t)k] = d[3 * k + c]; }
        HIP_TRY(hipMemcpy(r->q.ray_o[0] + (size_t)c * cap, so.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(r->q.ray_d[0] + (size_t)c * cap, sd.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    }
    uint32_t un = (uint32_t)n;
    HIP_TRY(hipMemcpy(r->scratch.p, &un, 4, hipMemcpyHostToDevice));
    Params flat = r->par; flat.nq = 1; flat.subcap = flat.cap;          // one flat queue for explicit rays
    hipLaunchKernelGGL(kExtend[r->trace_mode][0], dim3(grid_for((size_t)n, r->grid_trace, 1, r->trace_items)), dim3(r->trace_nt), r->lds_bytes, r->stream, r->scene->dev, flat, r->q, (Counters*)nullptr, 0,
                       (const uint32_t*)r->scratch.p, r->plan);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipMemcpy(prim_out, r->q.hit_prim, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t_out, r->q.hit_t, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (uv_out) {
        std::vector<float> u((size_t)n), v((size_t)n);
        HIP_TRY(hipMemcpy(u.data(), r->q.hit_u, (size_t)n * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(v.data(), r->q.hit_v, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < n; k++) { uv_out[2 * k] = u[(size_t)k]; uv_out[2 * k + 1] = v[(size_t)k]; }
    }
    return APT_OK;
}
APT_EXPORT int apt_occluded(apt_renderer* r, int32_t n, const float* o, const float* d, const float* tmax, int32_t* occ_out) {
    if (!r || n <= 0 || !o || !d || !tmax || !occ_out) return fail(APT_E_INVALID, "apt_occluded: bad argument");
    HIP_TRY(hipSetDevice(r->cfg.device));
    if (int rc = resolve_events(r)) return rc;
    std::vector<float> so((size_t)n * 3), sd((size_t)n * 3);
    for (int c = 0; c < 3; c++) for (int k = 0; k < n; k++) { so[(size_t)c * n + k] = o[3 * k + c]; sd[(size_t)c * n + k] = d[3 * k + c]; }
    DevBuf bo, bd, bt, bocc;
    HIP_TRY(upload(bo, so)); HIP_TRY(upload(bd, sd));
    std::vector<float> tm(tmax, tmax + n);
    HIP_TRY(upload(bt, tm)); HIP_TRY(bocc.alloc((size_t)n * 4));
    hipLaunchKernelGGL(kOccluded[r->trace_mode], dim3(grid_for((size_t)n, r->grid_shadow, 1, r->trace_items)), dim3(r->trace_nt), r->lds_bytes_any, r->stream, r->scene->dev, (uint32_t)n,
                       bo.as<float>(), bd.as<float>(), bt.as<float>(), bocc.as<int>(), r->plan);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipMemcpy(occ_out, bocc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return APT_OK;
}
APT_EXPORT int apt_rng_stream(int32_t device, uint32_t pixel, uint32_t seed, uint32_t sample, int32_t n, uint32_t* out) {
    if (n <= 0 || !out) return fail(APT_E_INVALID, "apt_rng_stream: bad argument");
    int ndev = 0;
    if (int rc = count_device(&ndev)) return rc;
    HIP_TRY(hipSetDevice(device));
    DevBuf b; HIP_TRY(b.alloc((size_t)n * 4));
    hipLaunchKernelGGL(k_rng_stream, dim3(1), dim3(64), 0, 0, pixel, seed, sample, n, b.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, b.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return APT_OK;
}

// BxDF / emitter probes: run the device shading code on explicit inputs (parity tests vs the golden vectors)
static void fill_bxdf(DevBxdf& b, const int32_t* bi, const float* bf) {
    memset(&b, 0, sizeof(b));
    b.type = bi[0]; b.is_delta = bi[1]; b.is_bsdf = bi[2];
    b.k_d = mk3(bf[0], bf[1], bf[2]); b.k_s = mk3(bf[3], bf[4], bf[5]); b.k_g = mk3(bf[6], bf[7], bf[8]); b.mean = mk3(bf[9], bf[10], bf[11]); b.ior = bf[12];
}
APT_EXPORT int apt_bxdf_probe(int32_t device, int32_t n, const int32_t* bxdf_i, const float* bxdf_f, const float* dirs12, float world_ior,
                              int32_t do_sample, uint32_t seed, float* out) {
    if (n <= 0 || !bxdf_i || !bxdf_f || !dirs12 || !out) return fail(APT_E_INVALID, "apt_bxdf_probe: bad argument");
    int ndev = 0;
    if (int rc = count_device(&ndev)) return rc;
    HIP_TRY(hipSetDevice(device));
    std::vector<DevBxdf> bx((size_t)n);
    for (int k = 0; k < n; k++) fill_bxdf(bx[(size_t)k], bxdf_i + 4 * k, bxdf_f + 13 * k);
    std::vector<float> in(dirs12, dirs12 + (size_t)n * 12);
    DevBuf dbx, din, dout;
    HIP_TRY(upload(dbx, bx)); HIP_TRY(upload(din, in));
    const size_t per = do_sample ? 9 : 4;
    HIP_TRY(dout.alloc((size_t)n * per * 4));
    if (do_sample) hipLaunchKernelGGL(k_bxdf_sample, dim3((n + 63) / 64), dim3(64), 0, 0, n, dbx.as<DevBxdf>(), din.as<float>(), world_ior, seed, dout.as<float>());
    else hipLaunchKernelGGL(k_bxdf_eval, dim3((n + 63) / 64), dim3(64), 0, 0, n, dbx.as<DevBxdf>(), din.as<float>(), world_ior, dout.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * per * 4, hipMemcpyDeviceToHost));
    return APT_OK;
}
APT_EXPORT int apt_medium_probe(int32_t device, int32_t n, const int32_t* med_i, const float* med_f, int32_t mode, const float* in7, uint32_t seed, float* out8) {
    if (n <= 0 || !med_i || !med_f || !in7 || !out8 || mode < 0 || mode > 2) return fail(APT_E_INVALID, "apt_medium_probe: bad argument");
    int ndev = 0;
    if (int rc = count_device(&ndev)) return rc;
    HIP_TRY(hipSetDevice(device));
    std::vector<DevMedium> md((size_t)n);
    for (int k = 0; k < n; k++) {
        const float* f = med_f + 16 * (size_t)k; DevMedium& m = md[(size_t)k]; memset(&m, 0, sizeof(m));
        m.type = med_i[k]; m.ior = f[0]; m.u_s = mk3(f[1], f[2], f[3]); m.u_a = mk3(f[4], f[5], f[6]); m.u_e = mk3(f[7], f[8], f[9]);
        m.par = mk3(f[10], f[11], f[12]); m.pdf = mk3(f[13], f[14], f[15]);
    }
    std::vector<float> in(in7, in7 + (size_t)n * 7);
    DevBuf dmed, din, dout;
    HIP_TRY(upload(dmed, md)); HIP_TRY(upload(din, in)); HIP_TRY(dout.alloc((size_t)n * 32));
    HIP_TRY(hipMemset(dout.p, 0, (size_t)n * 32));
    hipLaunchKernelGGL(k_medium_probe, dim3((n + 63) / 64), dim3(64), 0, 0, n, dmed.as<DevMedium>(), mode, din.as<float>(), seed, dout.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out8, dout.p, (size_t)n * 32, hipMemcpyDeviceToHost));
    return APT_OK;
}
APT_EXPORT int apt_emitter_probe(const apt_scene* sc, int32_t n, const float* in11, uint32_t seed, float* out12) {
    if (!sc || n <= 0 || !in11 || !out12) return fail(APT_E_INVALID, "apt_emitter_probe: bad argument");
    HIP_TRY(hipSetDevice(sc->device));
    std::vector<float> in(in11, in11 + (size_t)n * 11);
    DevBuf din, dout;
    HIP_TRY(upload(din, in)); HIP_TRY(dout.alloc((size_t)n * 48));
    hipLaunchKernelGGL(k_emitter_probe, dim3((n + 63) / 64), dim3(64), 0, 0, sc->dev, n, din.as<float>(), seed, dout.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out12, dout.p, (size_t)n * 48, hipMemcpyDeviceToHost));
    return APT_OK;
}
APT_EXPORT int apt_texture_probe(const apt_scene* sc, int32_t n, const int32_t* map_obj, const float* uv, float* out3) {
    if (!sc || n <= 0 || !map_obj || !uv || !out3) return fail(APT_E_INVALID, "apt_texture_probe: bad argument");
    if (!sc->dev.tex_i) return fail(APT_E_INVALID, "apt_texture_probe: the scene has no textures");
    for (int k = 0; k < n; k++) {
        const int m = map_obj[2 * k], o = map_obj[2 * k + 1];
        if (m < 0 || m > 2 || o < 0 || o >= sc->n_objects || !sc->dev.atlas[m]) return fail(APT_E_INVALID, "apt_texture_probe: no such texture");
    }
    HIP_TRY(hipSetDevice(sc->device));
    DevBuf dmo, duv, dout;
    std::vector<int> mo(map_obj, map_obj + 2 * (size_t)n); std::vector<float> vuv(uv, uv + 2 * (size_t)n);
    HIP_TRY(upload(dmo, mo)); HIP_TRY(upload(duv, vuv)); HIP_TRY(dout.alloc((size_t)n * 12));
    hipLaunchKernelGGL(k_texture_probe, dim3((n + 63) / 64), dim3(64), 0, 0, sc->dev, n, dmo.as<int>(), duv.as<float>(), dout.as<float>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out3, dout.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    return APT_OK;
}
// Shader clock while the whole chip is busy (bench.py prices its VALU roofline with it): median over the waves of a full grid of
// cycle-counter ticks per 100 MHz wall-clock tick.
APT_EXPORT int apt_measure_sclk_mhz(int32_t device, float* mhz) {
    if (!mhz) return fail(APT_E_INVALID, "apt_measure_sclk_mhz: bad argument");
    int ndev = 0;
    if (int rc = count_device(&ndev)) return rc;
    if (device < 0 || device >= ndev) return fail(APT_E_INVALID, "apt_measure_sclk_mhz: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int grid = std::max(1, prop.multiProcessorCount) * 8, n_waves = grid * (BLOCK / 64);
    DevBuf out, sink;
    HIP_TRY(out.alloc((size_t)n_waves * 16)); HIP_TRY(sink.alloc(16));
    for (int pass = 0; pass < 2; pass++) {            // the first pass warms the clocks up
        hipLaunchKernelGGL(k_clock_probe, dim3(grid), dim3(BLOCK), 0, 0, 20000, 1.0f, out.as<unsigned long long>(), sink.as<float>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h((size_t)n_waves * 2);
    HIP_TRY(hipMemcpy(h.data(), out.p, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> r;
    for (int w = 0; w < n_waves; w++) if (h[2 * (size_t)w + 1] > 0) r.push_back(100.0 * (double)h[2 * (size_t)w] / (double)h[2 * (size_t)w + 1]);
    if (r.empty()) return fail(APT_E_STATE, "apt_measure_sclk_mhz: the wall clock did not advance");
    std::nth_element(r.begin(), r.begin() + r.size() / 2, r.end());
    *mhz = (float)r[r.size() / 2];
    return APT_OK;
}
APT_EXPORT int apt_renderer_info(const apt_renderer* r, int32_t* spp_batch, int32_t* n_subqueues, int64_t* queue_bytes, int32_t* lds_bytes, const char** shade_variant, int32_t* trace_mode) {
    if (!r) return fail(APT_E_INVALID, "apt_renderer_info: null handle");
    if (spp_batch) *spp_batch = r->spp_batch;
    if (n_subqueues) *n_subqueues = r->nq;
    if (queue_bytes) *queue_bytes = (int64_t)r->pool.bytes * (int64_t)r->n_lanes;
    if (lds_bytes) *lds_bytes = (int32_t)r->lds_bytes;
    if (shade_variant) *shade_variant = r->shade_name.c_str();
    if (trace_mode) *trace_mode = r->trace_mode;
    return APT_OK;
}
