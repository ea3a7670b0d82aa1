// stages.hpp — the wavefront stage kernels of the gfx950 path tracer.
//
// Hot path being replaced: AdaPT's megakernel `Renderer.render`
// (renderer/vanilla_renderer.py:32-120: one launch per spp, one thread per pixel carrying a
// whole path through every bounce).  Here the same path-space computation is decomposed
// into stages that each stream dense SoA queues in HBM:
//
//   generate : camera ray per (pixel, sample) slot                       tracer_base.py:136-157
//   extend   : closest hit for every queued ray                          tracer_base.py:168-237 / path_tracer.py:338-394
//   shade    : emission + MIS, RR, NEE sampling -> shadow queue,         vanilla_renderer.py:44-117
//              BSDF sampling -> next ray queue (ballot-compacted)
//   shadow   : any-hit for every shadow ray; unoccluded ones add         tracer_base.py:239-278 / path_tracer.py:396-422
//              their contribution to the owning path's radiance
//   finalize : per pixel, sum the batch's samples in sample order,       vanilla_renderer.py:119-120
//              NaN -> 0, accumulate into the float3 framebuffer
//
// Queue organisation.  A batch holds P = owned_pixels * spp_batch paths.  Every queue is split
// into `nq` sub-queues (32 by default), each with its own region [q*subcap, (q+1)*subcap) and
// its own counters on private 128-byte lines.  Workgroup b serves sub-queue b % nq in every
// stage, and survivors of sub-queue q are appended to sub-queue q of the next queue, so
//   * queue-tail atomics are spread over nq addresses (one hot counter saturates at ~90
//     atomics/us on MI355X, which throttled the single-queue version), and
//   * with the dispatcher's observed block->XCD map (b % 8), a sub-queue is produced and
//     consumed under the same XCD's L2 in consecutive stages (speed only, never correctness).
// A path is identified for life by id = sample_in_batch * npix + local_pixel: the RNG is keyed
// by (global pixel, sample counter) and only a draw index travels with the path, and the
// radiance accumulator L[] is indexed by id, so results do not depend on queue position.
#pragma once
#include <hip/hip_runtime.h>

#include "rng.hpp"
#include "shading.hpp"
#include "traverse.hpp"
#include "vec.hpp"

// ------------------------------------------------------------- device views
struct DevMedium {            // bxdf/medium.py:71-78 + bxdf/phase.py:33-37 (volumetric path tracer only)
    int type;                 // -1 transparent, 0 hg, 1 multi-hg, 2 rayleigh, 3 mie (no phase function upstream)
    float ior;
    f3 u_s, u_a, u_e, par, pdf;
};
struct DevVolume {            // bxdf/volume.py:221-246 (grid volume of the volumetric tracer; type 0 = none, 2 = RGB)
    int type, xres, yres, zres;
    f3 albedo, trans, mini, maxi, majorant, pdf;
    f3 inv_r0, inv_r1, inv_r2; // rows of (rotation @ scale)^-1: world -> voxel coordinates
    DevMedium ph;             // phase function (type, par, pdf)
    const float* grid;        // zres*yres*xres*3 extinction per channel, [z][y][x][c]
};
struct DevScene {
    DevBvh bvh;
    SweepScene sweep;         // small scenes: uniform brute-force sweep instead of the BVH
    FlatScene flat;           // small scenes, fast build: precomputed-transform records, two per packed instruction (traverse.hpp "Flat sweep")
    const float* normals;     // n_prims*3
    const float4* vnormals;   // n_prims*3: the three vertex normals as 16-byte records (three loads per vertex; as nine packed floats they were nine 4-byte gathers, and the vector-memory pipe pays per lane address)
    const float* precom;      // n_prims*9  (v1-v0, v2-v0, v0) | sphere (centre, rrr, centre)
    const int* prim_obj;      // n_prims
    const int* prim_class;    // n_prims: material class of the owning object (sorted shading), see APT_CLASS_* in api.hip
    const int* obj_info;      // n_objects*3
    const int* emitter_id;    // n_objects
    const DevBxdf* bxdf;      // n_objects
    const DevSrc* src;        // n_sources
    int n_prims, n_objects, n_sources, has_vn;
    float world_ior;
    // image textures (tex_i == nullptr: none).  Maps: 0 albedo, 1 normal, 2 bump
    const float* uvs;         // n_prims*6
    const int* tex_i;         // n_objects*3*5: type, off_x, off_y, w, h
    const float* tex_f;       // n_objects*3*2: scale_u, scale_v
    const float* atlas[3];
    int atlas_w[3];
    // per-primitive shading record, 32 B: (n_g | sphere centre, object code = id or ~id for a sphere), (emitter id, k_d of the object's
    // material).  One fetch after the hit primitive is known replaces the chain primitive -> object -> {is-sphere flag -> normal,
    // emitter id, material}; the Lambertian-only kernel needs nothing else from the scene tables.
    const float4* prim_shade;
    const DevMedium* med;     // n_objects + 1 rows (the last one is the world's), nullptr when the scene declares no media
    DevVolume vol;
};

struct Params {
    float cam_r[9], cam_t[3];
    float inv_focal, half_w, half_h;
    int W, H, n_cols, npix;
    int band_width, rank, world;
    int do_crop, sx, ex, sy, ey;
    int max_bounce, S;
    float inv_S;
    int use_rr, use_mis, anti_alias, stratified, two_sides, rr_bounce_th;
    float rr_threshold;
    uint32_t seed;
    int cnt_base, spp_batch;
    uint32_t cap;             // nq * subcap: component stride of every path-indexed array
    uint32_t subcap;          // slots per sub-queue
    int nq;
    // A path's id word is (sample in batch << pix_bits) | local pixel: both halves come out with a shift and a mask
    // (no integer division per bounce); the radiance slot is sample * npix + local pixel.
    uint32_t pix_bits;
    const uint32_t* pix_key;  // local pixel -> global pixel index x * H + y = the RNG key (host-built; band mapping folded in)
    float inv_ns, inv_ns1;    // 1 / n_sources, 1 / (n_sources - 1): sample_light's pdfs (path_tracer.py:537-554)
    int fix_par;              // flat sweep: queue parity of the current bounce (which shadow fix-up list the hot shadow kernel appends to); set per launch
    int volumetric_flat;      // 1: volumetric render (the flat shadow kernels never run: the fix-up launch has no shadow list)
    int nee_vm;               // 1: light samples are queued BY VERTEX (one slot per vertex and sub-queue, sample s in plane s of the sub-queue's region, unwanted samples marked tmax < 0), so that the flat shadow kernel adds a vertex's samples with ONE read-modify-write and the shade kernel moves the queue tail once per tile row (flat sweep, S > 1)
    int l_planes;             // radiance planes of L: light sample s of a vertex adds into plane s (2 <= S <= 4), so that no two entries of a shadow launch share a slot; 1 otherwise
    int fused;                // 2: the shade kernel traces its own light sample and continuation ray, k_generate its camera rays (flat sweep, one sample per vertex, unsorted: shade_stage.hpp "rays traced in place"):
                              //    no extend, shadow or fix-up launch per bounce, no shadow queue, radiance travels with the path's record, rays that hit nothing never enter a queue; 0: staged
    float w_min[3], w_max[3]; // world box = (objects U camera) +- 0.1 (path_tracer.py:130-138); volumetric tracer only
};

// SoA queues; every array has `cap` (shadow: sh_cap) entries per component
struct Queues {
    float* ray_o[2]; float* ray_d[2];           // 3 components each
    float* hit_t; int* hit_prim; float* hit_u; float* hit_v;
    // rays traced in place (Params::fused == 2): the path record as four 16-byte planes per queue parity - A = (ray origin, hit distance),
    // B = (ray direction, pm), C = (throughput, path id), D = (radiance so far, pdf of the ray) - and, where somebody reads them, the hit's
    // barycentrics; pm = hit primitive (8 bits, TR_NO_PRIM: nothing hit) | draw index << 8 | specular bit << 24.  Still SoA - a wave's 64
    // entries of a plane are 1 KiB in a row - but a record is 4 loads and 4 stores instead of 16 and 18, and 8 base pointers instead of 30:
    // the shade kernel's scalar registers no longer overflow into VGPR lanes (218 v_readlane per tile row before).  Null elsewhere.
    // Every plane holds tr_ncls + 1 queues of `cap` slots each: one per material class (class-sorted shading: the shade kernel that traces a
    // continuation ray appends the record to the queue of the HIT primitive's class - sorting costs no pass of its own) and a staging queue
    // (index tr_ncls) for the rays whose hit is left to the reference-order code; slot = queue * cap + sub-queue * subcap + position.
    // tr_stage_top (the shipped one-queue case): no memory for the staging queue - its entries grow DOWN from the top of the sub-queue's own
    // region (slot = sub-queue * subcap + subcap - 1 - position) while the queue grows up from the bottom; the two cannot meet, because what
    // a bounce appends is at most what it read.
    float4* tr[2][4]; float2* tr_uv[2]; int tr_ncls, tr_stage_top;
    uint32_t* fix_ext; uint32_t* fix_sh;         // flat sweep: fix-up lists, sub-queue-local entry indices (null elsewhere)
    float* thr[2]; uint32_t* id[2]; uint32_t* meta[2]; float* pdf[2];
    float* sh_o; float* sh_d; float* sh_tmax; float* sh_c; uint32_t* sh_id;
    uint32_t* sh_walk[2];                        // volumetric, scenes with null surfaces: slot lists of the samples that walk on (ping-pong)
    float* L;                                    // 3 components, indexed by path id
    uint32_t sh_cap, sh_subcap;
    // Material-sorted shading (scenes with >= 2 material classes): extend drops misses and appends each hit path's full record (ray + state
    // + hit, 64 B) to the dense queue of its material class, so every shade launch runs one specialised kernel over coherent waves; the
    // volumetric tracer's EVENT queues (volumetric.hpp k_vevent) are the same thing.  n_classes == 0: unsorted (single-class scenes).
    // The queues are FOUR 16-byte planes: A = (ray origin, hit distance), B = (ray direction, hit primitive), C = (throughput, path id),
    // D = (meta, pdf, u, v); every plane holds n_classes queues of `cap` slots, slot = class * cap + sub-queue * subcap + position.  A
    // record is 4 stores and 4 loads, and - what matters in the BVH walk's hand-in, where a wave's finished rays belong to several classes -
    // the class is part of the lane's OFFSET: no loop over the classes present, no per-class queue pointers.
    float4* cq[4];
    int n_classes;
};
#define APT_MAX_CLASSES 8
// what one shade launch reads: either ray queue `cur` + the hit arrays (unsorted) or one class queue (sorted)
struct ShadeIn {
    const float* ray_o; const float* ray_d; const float* thr; const uint32_t* id; const uint32_t* meta; const float* pdf;
    const float* t; const int* prim; const float* u; const float* v;
    const uint32_t* counts;                      // per-sub-queue entry counts (stride CNT_PAD)
    int cls;                                     // class-sorted renders: which of the packed class queues (Queues::cq) this launch reads
};

#ifndef APT_MAX_NQ
#define APT_MAX_NQ 32
#endif
#define CNT_PAD 32                               // one counter per 128-byte line
enum { ST_SAMPLES = 0, ST_EXTEND, ST_SHADE, ST_SHADOW, ST_SHADOW_TRACED, ST_LIT, ST_DRAWS, ST_POISON, ST_TRACK, ST_OVERFLOW, ST_COUNT };      // ST_OVERFLOW: volumetric paths whose draw index left its 23 bits
struct Counters {
    uint32_t n_active[2][APT_MAX_NQ * CNT_PAD];
    uint32_t n_shadow[APT_MAX_NQ * CNT_PAD];
    uint32_t n_walk[8][APT_MAX_NQ * CNT_PAD];     // volumetric: light samples still walking after pass p-1 (pass p reads list p, fills list p + 1)
    uint32_t n_cls[8][APT_MAX_NQ * CNT_PAD];
    uint32_t n_fix_ext[2][APT_MAX_NQ * CNT_PAD];  // flat sweep: entries handed to the fix-up launch of the extend stage (by queue parity) ...
    uint32_t n_fix_sh[2][APT_MAX_NQ * CNT_PAD];   // ... and of the shadow stage, by the parity of the bounce that listed them (stages.hpp "fix-up lists")
    uint32_t n_work[2][APT_MAX_NQ * CNT_PAD];     // BVH walk with dynamic fetch: next unclaimed queue position (0 extend, 1 shadow); zeroed with the batch's counters, then by the other walk kernel (surface renders with light samples) or by the host before the launch
    uint32_t n_tr[3][8 + 1][APT_MAX_NQ * CNT_PAD];   // rays traced in place: entries of the queues (material classes, then the staging queue) that bounce k of the batch reads, at [k % 3] (bounce k appends to [(k + 1) % 3] and zeroes [(k + 2) % 3], which bounce k - 1 read: no launch in between has to reset a counter)
    uint32_t fix_claim[APT_MAX_NQ * CNT_PAD];     // rays traced in place: which launch of the batch (1 + bounce) has had its fix-up lists claimed by a wave ...
    uint32_t fix_done[APT_MAX_NQ * CNT_PAD];      // ... and served (shade_stage.hpp fix_prologue)
    unsigned long long stats[APT_MAX_NQ][16];    // [q][ST_*], 128 bytes per sub-queue
#ifdef APT_WALK_STATS
    unsigned long long wdbg[2][8];               // walk scheduling, [closest-hit | any-hit]: node iterations, lanes in them, primitive iterations, lanes in them, lanes holding a ray (summed over iterations), refills, lanes claimed, -
#endif
};

// meta word: draw index [0,16) | bounce [16,24) | is_specular bit 24
APT_D uint32_t pack_meta(uint32_t draw, uint32_t bounce, bool spec) { return (draw & 0xffffu) | ((bounce & 0xffu) << 16) | (spec ? (1u << 24) : 0u); }

#define BLOCK 256
#define TR_NO_PRIM 0xffu
APT_D uint32_t tr_pack(int prim, uint32_t draw, bool spec) { return (prim < 0 ? TR_NO_PRIM : (uint32_t)prim) | ((draw & 0xffffu) << 8) | (spec ? (1u << 24) : 0u); }
APT_D int tr_prim(uint32_t pm) { return ((pm & 0xffu) == TR_NO_PRIM) ? -1 : (int)(pm & 0xffu); }
APT_D uint32_t tr_meta(uint32_t pm, uint32_t bounce) { return pack_meta(pm >> 8, bounce, ((pm >> 24) & 1u) != 0u); }      // the staged pipeline's meta word
// One light sample per vertex, or one radiance plane per light sample (Params::l_planes): no two entries of a shadow launch add into the
// same slot and the adds are plain read-modify-writes.  Measured on C3 (S = 4) with float atomics instead: k_shadow 17 % VALU-busy, 3.9x
// its algorithmic HBM writes (every atomic is an L2 read-modify-write of a sector), and run-to-run differences in the last bit.
#ifndef APT_EXCLUSIVE_L
#define APT_EXCLUSIVE_L(p) ((p).S == 1 || (p).l_planes == (p).S)
#endif

// LDS of the BVH-walk stages (dynamic, sized per scene by the host): the traversal stack, stack_depth * BLOCK 8-byte groups laid out [level][lane]
struct LdsPlan { int stack_depth; uint2* ovf; int ovf_stride; };     // stack_depth: levels kept in LDS; ovf: global spill columns (one per thread of the largest grid)
extern __shared__ float4 s_dyn[];
typedef __attribute__((address_space(3))) float lds_f;
APT_D TravStack make_stack(const LdsPlan& plan) {
    TravStack ts;
    ts.lds = (lds_u2*)(reinterpret_cast<grp_t*>(s_dyn)) + threadIdx.x;
    ts.stride = BLOCK; ts.k = plan.stack_depth; ts.ovf_stride = plan.ovf_stride;
    ts.ovf = plan.ovf ? (glb_u2*)(reinterpret_cast<grp_t*>(plan.ovf) + (blockIdx.x * blockDim.x + threadIdx.x)) : (glb_u2*)nullptr;
    ts.lut = (lds_u8*)nullptr;
    return ts;
}
// the walk kernels' stack: + the priority-permutation table behind the stack and the parked path state (api.hip sizes the LDS: stack, 6 floats per thread, 2 KiB)
APT_D TravStack make_walk_stack(const LdsPlan& plan) {
    TravStack ts = make_stack(plan);
    ts.lut = (lds_u8*)(reinterpret_cast<float*>(s_dyn) + (size_t)plan.stack_depth * BLOCK * 2 + 6 * BLOCK);
    fill_permute_lut(ts.lut);
    __syncthreads();
    return ts;
}

APT_D uint32_t lane_id() { return threadIdx.x & 63u; }
// Radiance slots L: one float4 (r, g, b, -) per path and radiance plane, addressed by the slot word `l_off` = slot index << 2 (its low two
// bits carry the plane in a shadow-queue entry).  AoS on purpose: these are the only accesses keyed by PATH ID instead of queue position -
// scattered once queues are compacted and sorted - and as three 4-byte accesses to three component planes every read-modify-write touched
// three sectors; as one 16-byte access it touches one.
// (Planes are whole arrays, plane p at L + p * 4 * cap: keeping a path's planes next to each other in one 64-byte line measured the same or slightly worse, C3 shadow 17.5 -> 19.0 ms per 128 spp.)
APT_D float* L_slot(float* L, uint32_t cap, uint32_t code) { return L + (size_t)(code & 3u) * 4 * cap + (size_t)(code & ~3u); }      // (slot << 2 floats = 16 bytes per slot)
APT_D f3 ldL(float* L, uint32_t cap, uint32_t code) { const float4 v = *reinterpret_cast<const float4*>(L_slot(L, cap, code)); return mk3(v.x, v.y, v.z); }
APT_D void stL(float* L, uint32_t cap, uint32_t code, f3 v) { *reinterpret_cast<float4*>(L_slot(L, cap, code)) = make_float4(v.x, v.y, v.z, 0.f); }
// A light sample's contribution into its path's radiance slot.  With ONE light sample per
// path vertex (p.S == 1) no two entries of a shadow launch share a slot and nothing else writes L while the launch runs, so the add is
// a plain read-modify-write: deterministic, and not an L2 atomic per component.  With several samples per vertex the entries of one
// path sit in different waves: each sample adds into its own radiance plane (Params::l_planes; 2 <= S <= 4), or, beyond four samples, float atomics.
APT_D void add_radiance(float* L, uint32_t cap, uint32_t code, f3 c, bool exclusive) {
    if (exclusive) { const f3 a = ldL(L, cap, code); stL(L, cap, code, mk3(a.x + c.x, a.y + c.y, a.z + c.z)); }
    else { float* p_ = L_slot(L, cap, code); atomicAdd(p_, c.x); atomicAdd(p_ + 1, c.y); atomicAdd(p_ + 2, c.z); }
}
// number of set bits of a ballot mask below this lane (v_mbcnt: no lane-mask registers to keep alive)
APT_D uint32_t rank_in(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
// append `flag` lanes of the wave to the queue counted by *counter; returns this lane's position
APT_D uint32_t wave_append(bool flag, uint32_t* counter) {
    unsigned long long m = __ballot(flag);
    uint32_t base = 0;
    if (lane_id() == 0 && m) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);      // v_readlane: lane 0's value as a scalar (a shuffle would go through the LDS crossbar)
    return base + rank_in(m);
}

// The same in two halves, for appends whose position is needed later than the flag is known: append_issue() sends the tail atomic on its
// way, append_pos() waits for it.  (vmcnt counts in order: waiting for a returning atomic also waits for every load issued before it.)
struct Append { unsigned long long m; uint32_t raw; };
APT_D Append append_issue(bool flag, uint32_t* counter) {
    Append a; a.m = __ballot(flag); a.raw = 0;
    if (lane_id() == 0 && a.m) a.raw = atomicAdd(counter, (uint32_t)__popcll(a.m));
    return a;
}
APT_D uint32_t append_pos(const Append& a) { return (uint32_t)__builtin_amdgcn_readlane((int)a.raw, 0) + rank_in(a.m); }

// the same for blocks of `k` consecutive entries per flagged lane; returns the position of this lane's block
APT_D uint32_t wave_append_n(bool flag, uint32_t* counter, uint32_t k) {
    unsigned long long m = __ballot(flag);
    uint32_t base = 0;
    if (lane_id() == 0 && m) base = atomicAdd(counter, (uint32_t)__popcll(m) * k);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);      // v_readlane: lane 0's value as a scalar (a shuffle would go through the LDS crossbar)
    return base + rank_in(m) * k;
}

// Append to one of several queues at once (rays traced in place: material classes + the staging queue, `nq_` of them; which = this lane's
// queue, -1: none): the tails of all queues move with ONE atomic instruction (lane c carries queue c's count), as in k_extend's class
// appends.  tr_append_issue sends it on its way, tr_append_pos waits for it.  `counters` = the first queue's tail of this sub-queue, queue c's
// is `stride` words further per c.
struct TrAppend { uint32_t raw, rank; };
APT_D TrAppend tr_append_issue(int which, int nq_, uint32_t* counters, uint32_t stride) {
    TrAppend a; a.raw = 0u; a.rank = 0u;
    uint32_t cnt_vec = 0u;
    for (int c = 0; c < nq_; c++) {
        const unsigned long long m = __ballot(which == c);
        if (which == c) a.rank = rank_in(m);
        if ((int)lane_id() == c) cnt_vec = (uint32_t)__popcll(m);
    }
    if ((int)lane_id() < nq_ && cnt_vec) a.raw = atomicAdd(counters + (size_t)lane_id() * stride, cnt_vec);
    return a;
}
APT_D uint32_t tr_append_pos(const TrAppend& a, int which) { return (uint32_t)__shfl((int)a.raw, which < 0 ? 0 : which) + a.rank; }

// one record of a packed class queue (Queues::cq)
APT_D void cq_store(const Queues& q, uint32_t slot, f3 o, f3 d, f3 thr, uint32_t id, uint32_t meta, float pdf, float t, int prim, float u, float v) {
    const uint32_t so = slot << 4;
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(q.cq[0]) + so) = make_float4(o.x, o.y, o.z, t);
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(q.cq[1]) + so) = make_float4(d.x, d.y, d.z, __int_as_float(prim));
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(q.cq[2]) + so) = make_float4(thr.x, thr.y, thr.z, __uint_as_float(id));
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(q.cq[3]) + so) = make_float4(__uint_as_float(meta), pdf, u, v);
}

// Queue addressing.  Every queue array is indexed by a 32-bit slot whose BYTE offset also fits 32 bits (the host
// refuses batches with 12 * capacity >= 4 GiB), and every base pointer is wave-uniform.  Written as
// `uniform base + zero-extended 32-bit byte offset` the backend selects the SGPR-base form
// `global_load_dword v, v_off, s[base:base+1]`: one VGPR offset serves all arrays of a record and the component
// strides are added on the scalar unit.  With `ptr[index]` each access costs a 64-bit VALU address and a VGPR pair.
template <typename T> APT_D T ldq(const T* base, uint32_t off) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off); }
template <typename T> APT_D void stq(T* base, uint32_t off, T v) { *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off) = v; }
APT_D f3 ld3q(const float* base, uint32_t stride, uint32_t off) { return mk3(ldq(base, off), ldq(base + stride, off), ldq(base + 2 * stride, off)); }
APT_D void st3q(float* base, uint32_t stride, uint32_t off, f3 v) { stq(base, off, v.x); stq(base + stride, off, v.y); stq(base + 2 * stride, off, v.z); }
// wave-uniform tally: how many lanes of the wave have `flag` set (lives in an SGPR)
APT_D uint32_t wave_count(bool flag) { return (uint32_t)__popcll(__ballot(flag)); }
APT_D void flush_uniform(uint32_t v, unsigned long long* counter) { if (lane_id() == 0 && v) atomicAdd(counter, (unsigned long long)v); }
// end-of-kernel statistics: per-lane register tallies -> one atomic per wave per counter
APT_D void flush_stat(uint32_t v, unsigned long long* counter) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane_id() == 0 && v) atomicAdd(counter, (unsigned long long)v);
}

// An emitter record through the constant address space: with a wave-uniform address (scenes with one light) the backend emits
// scalar loads into SGPRs, outside the light-sample loop, instead of a 64-byte per-lane fetch at the head of every sample's
// dependent chain (record -> triangle pick -> vertices).
APT_D DevSrc ld_src_uniform(const DevSrc* p) {
    const ci_ptr w = (ci_ptr)reinterpret_cast<const int*>(p);
    DevSrc s; int* dst = reinterpret_cast<int*>(&s);
    for (int k = 0; k < (int)(sizeof(DevSrc) / 4); k++) dst[k] = w[k];
    return s;
}

// An emitter record picked per lane (scenes with several lights): the 64-byte record as FOUR 16-byte loads.  Left to the compiler the struct
// copy dissolves into one 4-byte load per field at its use site - up to nine scattered loads per light sample, most of the scattered
// accesses a vertex of the mesh scenes makes (C4: 18 of 25 with two light samples) - and the vector-memory pipe pays per lane address.
#ifndef APT_SRC_VEC
#define APT_SRC_VEC 1
#endif
APT_D DevSrc ld_src_lane(const DevSrc* p) {
#if APT_SRC_VEC
    static_assert(sizeof(DevSrc) == 64, "DevSrc is read as four float4");
    const float4* q4 = reinterpret_cast<const float4*>(p);
    float4 a = q4[0], b = q4[1], c = q4[2], d = q4[3];
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));      // (keeps the four loads whole)
    asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w), "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
    DevSrc s;
    s.type = __float_as_int(a.x); s.bool_bits = __float_as_int(a.y); s.obj_ref_id = __float_as_int(a.z); s.prim_first = __float_as_int(a.w);
    s.intensity = mk3(b.x, b.y, b.z); s.dir = mk3(b.w, c.x, c.y); s.pos = mk3(c.z, c.w, d.x);
    s.inv_area = d.y; s.r = d.z; s.prim_count = __float_as_int(d.w);
    return s;
#else
    return *p;
#endif
}

// ... and an object's surface-model record (80 bytes) as five
#ifndef APT_BXDF_VEC
#define APT_BXDF_VEC 1
#endif
APT_D DevBxdf ld_bxdf_lane(const DevBxdf* p) {
#if APT_BXDF_VEC
    static_assert(sizeof(DevBxdf) == 80, "DevBxdf is read as five float4");
    const float4* q4 = reinterpret_cast<const float4*>(p);
    float4 a = q4[0], b = q4[1], c = q4[2], d = q4[3], e = q4[4];
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
    asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w), "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w), "+v"(e.x));
    DevBxdf x;
    x.type = __float_as_int(a.x); x.is_delta = __float_as_int(a.y); x.is_bsdf = __float_as_int(a.z); x._pad = 0;
    x.k_d = mk3(b.x, b.y, b.z); x.k_s = mk3(b.w, c.x, c.y); x.k_g = mk3(c.z, c.w, d.x); x.mean = mk3(d.y, d.z, d.w);
    x.ior = e.x; x._pad2[0] = x._pad2[1] = x._pad2[2] = 0.f;
    return x;
#else
    return *p;
#endif
}

// local pixel -> (global column, row)
APT_D void local_to_global(const Params& p, uint32_t lp, int& i, int& j) {
    int lc = (int)(lp / (uint32_t)p.H);
    j = (int)(lp % (uint32_t)p.H);
    int lb = lc / p.band_width, w = lc % p.band_width;
    i = (lb * p.world + p.rank) * p.band_width + w;
}

// workgroup -> (sub-queue, first slot, stride) of the persistent loop over a sub-queue
struct SubLoop { int q; uint32_t first, stride; };
APT_D SubLoop sub_loop(int nq, int nt = BLOCK) {
    SubLoop s;
    s.q = (int)(blockIdx.x % (uint32_t)nq);
    s.first = (blockIdx.x / (uint32_t)nq) * (uint32_t)nt;
    s.stride = (gridDim.x / (uint32_t)nq) * (uint32_t)nt;
    return s;
}
// trace kernels: MODE 0 BVH walk, 1 wave/workgroup sweep, 2 tiled sweep (its own, larger workgroup), 3 flat sweep (fast build only)
#ifndef APT_TILE_NT
#define APT_TILE_NT 512
#endif
// occupancy targets (waves per SIMD the register allocator must allow): the stages are latency-bound on dependent
// table lookups and LDS round trips, so more resident waves beat a few spilled registers (measured)
#ifndef APT_TILE_WAVES
#define APT_TILE_WAVES 6
#endif
#define TRACE_NT(MODE) ((MODE) == 2 ? APT_TILE_NT : BLOCK)

// ----------------------------------------------------------------- generate
// wave w of the id space feeds sub-queue w % nq at position (w / nq) * 64 + lane: dense and
// atomic-free unless a crop window makes some lanes inactive.
// TRACE (rays traced in place, Params::fused == 2): the camera ray meets the scene's records here and the entry carries its hit record
// (Queues::tr, parity 0, the queue of the hit primitive's class; counted in n_tr[0]); a ray that hits nothing is not queued at all.
template <bool TRACE>
APT_D void generate_body(const DevScene* sc, const Params& p, const Queues& q, Counters* cnt) {
    const uint32_t total = (uint32_t)p.npix * (uint32_t)p.spp_batch;
    const uint32_t n_waves = (total + 63u) / 64u;
    const uint32_t wave_stride = gridDim.x * (BLOCK / 64);
    uint32_t t_samples = 0, t_draws = 0;
    for (uint32_t w = blockIdx.x * (BLOCK / 64) + threadIdx.x / 64u; w < n_waves; w += wave_stride) {
        const uint32_t idx = w * 64u + lane_id();
        const int sq = (int)(w % (uint32_t)p.nq);
        bool valid = idx < total, alive = false;
        f3 dir = mk3(0.f, 0.f, 1.f);
        uint32_t draws = 0;
        if (valid) {
            uint32_t lp = idx % (uint32_t)p.npix, s = idx / (uint32_t)p.npix;
            int i, j; local_to_global(p, lp, i, j);
            for (int pl = 0; pl < p.l_planes; pl++) stL(q.L, p.cap, (idx << 2) | (uint32_t)pl, splat3(0.f));
            alive = !p.do_crop || (i >= p.sx && i < p.ex && j >= p.sy && j < p.ey);
            if (alive) {
                int sample_cnt = p.cnt_base + (int)s + 1;        // cnt is incremented before the pixel loop
                Philox rng; rng_init(rng, (uint32_t)(i * p.H + j), p.seed, (uint32_t)sample_cnt, 0u);
                float vx = 0.5f, vy = 0.5f;
                if (p.anti_alias) {
                    if (p.stratified) {
                        int mod_val = pymod(sample_cnt, 16);
                        vx = (float)(mod_val % 4) * 0.25f + rng_float(rng) * 0.25f;
                        vy = (float)(mod_val / 4) * 0.25f + rng_float(rng) * 0.25f;
                    } else {
                        const float eps = 1e-4f, inv_eps = (float)(1 - 1e-4 * 2.);
                        vx = rng_float(rng) * inv_eps + eps;
                        vy = rng_float(rng) * inv_eps + eps;
                    }
                }
                f3 cd = mk3((p.half_w + vx - (float)i) * p.inv_focal, ((float)j - p.half_h - vy) * p.inv_focal, 1.f);
                m33 R;
                for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) R.m[a][b] = p.cam_r[3 * a + b];
                dir = normalize(mul(R, cd));
                draws = rng.draw;
            }
        }
        uint32_t pos;
        uint32_t* q0_counter = &cnt->n_active[0][sq * CNT_PAD];
        if (TRACE) pos = 0;
        else if (p.do_crop) pos = wave_append(alive, q0_counter);
        else {
            pos = (w / (uint32_t)p.nq) * 64u + lane_id();
            unsigned long long m = __ballot(alive);
            if (lane_id() == 0) atomicAdd(q0_counter, (uint32_t)__popcll(m));   // nq-way spread, ordered by w
        }
#if APT_FAST
        if (TRACE) {
            const f3 cam_o = mk3(p.cam_t[0], p.cam_t[1], p.cam_t[2]);
            float tr_t = 0.f; int tr_run = -1, tr_idx = -1;
            if (__any(alive)) tr_idx = flat_closest1(sc->flat, cam_o, dir, 1e7f, tr_t, tr_run);
            const bool defer = alive && (tr_run >= 0 || flat_needs_cull(sc->flat, dir));
            HitRec hr; hr.t = 1e7f; hr.prim = -1; hr.u = hr.v = 0.f;
            int hit_cls = 0;
            if (alive && !defer && tr_idx >= 0) flat_resolve(sc->flat, tr_idx, tr_t, cam_o, dir, hr, hit_cls);
            // the entry joins the queue of the hit primitive's class (staging queue: hit left to the next launch's prologue); a camera ray that hits nothing ends here
            const int ocls = !alive ? -1 : (defer ? q.tr_ncls : (hr.prim >= 0 ? (q.tr_ncls > 1 ? hit_cls : 0) : -1));
            const TrAppend app = tr_append_issue(ocls, q.tr_ncls + 1, &cnt->n_tr[0][0][sq * CNT_PAD], APT_MAX_NQ * CNT_PAD);
            pos = tr_append_pos(app, ocls);
            if (alive) t_samples++;
            if (ocls >= 0) {
                const uint32_t slot = (q.tr_stage_top && ocls == q.tr_ncls) ? (uint32_t)sq * p.subcap + p.subcap - 1u - pos : (uint32_t)ocls * p.cap + (uint32_t)sq * p.subcap + pos, so = slot << 4;
                stq(q.tr[0][0], so, make_float4(cam_o.x, cam_o.y, cam_o.z, hr.t));
                stq(q.tr[0][1], so, make_float4(dir.x, dir.y, dir.z, __uint_as_float(tr_pack(hr.prim, draws, false))));
                stq(q.tr[0][2], so, make_float4(1.f, 1.f, 1.f, __uint_as_float(((idx / (uint32_t)p.npix) << p.pix_bits) | (idx % (uint32_t)p.npix))));
                stq(q.tr[0][3], so, make_float4(0.f, 0.f, 0.f, 1.f));
                if (sc->has_vn || sc->tex_i != nullptr) { float2 uv_; uv_.x = hr.u; uv_.y = hr.v; stq(q.tr_uv[0], slot << 3, uv_); }
            }
        }
#endif
        if (alive && !TRACE) {
            const uint32_t so = ((uint32_t)sq * p.subcap + pos) << 2;
            st3q(q.ray_o[0], p.cap, so, mk3(p.cam_t[0], p.cam_t[1], p.cam_t[2]));
            st3q(q.ray_d[0], p.cap, so, dir);
            st3q(q.thr[0], p.cap, so, splat3(1.f));
            stq(q.id[0], so, ((idx / (uint32_t)p.npix) << p.pix_bits) | (idx % (uint32_t)p.npix));
            stq(q.meta[0], so, pack_meta(draws, 0u, false));
            stq(q.pdf[0], so, 1.f);
            t_samples++;
        }
        t_draws += draws;
    }
    const int sq0 = (int)((blockIdx.x * (BLOCK / 64) + threadIdx.x / 64u) % (uint32_t)p.nq);
    flush_stat(t_samples, &cnt->stats[sq0][ST_SAMPLES]);
    if (TRACE) flush_stat(t_samples, &cnt->stats[sq0][ST_EXTEND]);
    flush_stat(t_draws, &cnt->stats[sq0][ST_DRAWS]);
}
__global__ void __launch_bounds__(BLOCK) k_generate(Params p, Queues q, Counters* cnt) { generate_body<false>(nullptr, p, q, cnt); }
#if APT_FAST
__global__ void __launch_bounds__(BLOCK) k_generate_trace(DevScene sc, Params p, Queues q, Counters* cnt) { generate_body<true>(&sc, p, q, cnt); }
#endif

// ------------------------------------------------------------------- extend
// closest hit for ray queue `cur`.  Also recycles the counters nobody reads any more: the
// next-ray queue of this bounce (it was the current queue of the previous bounce) and the
// shadow queue.  `n_src` = per-sub-queue counts (normally cnt->n_active[cur]).
// MODE 0: BVH traversal (LDS-staged nodes + per-lane LDS stack); MODE 1: wave-uniform sweep (small scenes)
template <int MODE, int SORTED>
__global__ void __launch_bounds__(TRACE_NT(MODE), (MODE == 2 ? APT_TILE_WAVES : 1)) k_extend(DevScene sc, Params p, Queues q, Counters* cnt, int cur, const uint32_t* n_src, LdsPlan plan) {
    __shared__ float s_sweep[MODE == 1 ? APT_SWEEP_LDS_FLOATS(BLOCK) : 1];
    const SubLoop sl = sub_loop(p.nq, TRACE_NT(MODE));
    const uint32_t n = n_src[sl.q * CNT_PAD];
    if (cnt && sl.first == 0 && threadIdx.x == 0) {
        cnt->n_shadow[sl.q * CNT_PAD] = 0; cnt->n_active[cur ^ 1][sl.q * CNT_PAD] = 0;
        for (int w = 0; w < 8; w++) cnt->n_walk[w][sl.q * CNT_PAD] = 0;
        cnt->stats[sl.q][ST_EXTEND] += n;
    }
    const float* ro = q.ray_o[cur]; const float* rd = q.ray_d[cur];
    const uint32_t qbase = (uint32_t)sl.q * p.subcap;
    for (uint32_t base = sl.first; base < n; base += sl.stride) {
        const uint32_t pos = base + threadIdx.x;
        const bool valid = pos < n;
        const uint32_t idx = qbase + (valid ? pos : n - 1);            // idle lanes re-read the last ray (never written back)
        const uint32_t io = idx << 2;
        const f3 o = ld3q(ro, p.cap, io);
        const f3 d = ld3q(rd, p.cap, io);
        HitRec rec; rec.t = 1e7f; rec.prim = -1; rec.u = 0.f; rec.v = 0.f;
        if (MODE == 0) traverse<false>(sc.bvh, make_stack(plan), o, d, rec);
        else if (MODE == 1) sweep_wg<false, BLOCK>(sc.sweep, o, d, rec, valid, s_sweep);
        else sweep_tile<false, APT_TILE_NT>(sc.sweep, o, d, rec, valid, reinterpret_cast<float*>(s_dyn));
        if (!SORTED) {
            if (valid) { stq(q.hit_t, io, rec.t); stq(q.hit_prim, io, rec.prim); stq(q.hit_u, io, rec.u); stq(q.hit_v, io, rec.v); }
        } else {
            // sort by material class: misses vanish here, every hit path's record goes to the dense queue of its class.  The queue tails
            // of ALL classes move with one atomic instruction (lane c carries class c's count), so a tile pays one memory round trip for
            // its appends instead of one per class present; the record stores then run class by class with scalar queue pointers.
            // the rest of the path's record travels with the hit: requested before the class lookup and the tail atomic, so that all three round trips overlap
            const f3 st_thr = ld3q(q.thr[cur], p.cap, io); const uint32_t st_id = ldq(q.id[cur], io), st_meta = ldq(q.meta[cur], io); const float st_pdf = ldq(q.pdf[cur], io);
            const int cls = (valid && rec.prim >= 0) ? sc.prim_class[rec.prim] : -1;
            uint32_t my_rank = 0, cnt_vec = 0;
            for (int c = 0; c < q.n_classes; c++) {
                const unsigned long long m = __ballot(cls == c);
                if (cls == c) my_rank = rank_in(m);
                if ((int)lane_id() == c) cnt_vec = (uint32_t)__popcll(m);
            }
            uint32_t tail = 0;
            if ((int)lane_id() < q.n_classes && cnt_vec) tail = atomicAdd(&cnt->n_cls[lane_id()][sl.q * CNT_PAD], cnt_vec);
            const uint32_t cpos = (uint32_t)__shfl((int)tail, cls < 0 ? 0 : cls) + my_rank;
            if (cls >= 0) cq_store(q, (uint32_t)cls * p.cap + qbase + cpos, o, d, st_thr, st_id, st_meta, st_pdf, rec.t, rec.prim, rec.u, rec.v);
        }
    }
}



// ------------------------------------------------------- extend, BVH walk with dynamic ray fetch
// Incoherent rays need very different numbers of traversal steps, and in the plain persistent loop a wave is as slow as its slowest
// ray: lanes that have finished idle until the last one is done.  Here a wave keeps walking only while at least APT_DYN_MIN_ACTIVE
// of its lanes still hold a ray; when fewer do, the finished lanes hand in their results and claim fresh rays from the sub-queue's
// work counter (one atomic per wave), so the walk loops always run with a mostly full wave (Aila & Laine's "persistent threads with
// dynamic fetch", re-cut for 64-wide waves and an LDS stack).  Per-ray arithmetic and the visiting order inside a ray are those of
// traverse<false>; only the assignment of rays to lanes changes, and hits are written to the ray's own slot.
#ifndef APT_DYN_MIN_ACTIVE_SH
#define APT_DYN_MIN_ACTIVE_SH 32   // the any-hit walk's threshold (its hand-in is one radiance add; measured below)
#endif
#ifndef APT_DYN_MIN_ACTIVE
#define APT_DYN_MIN_ACTIVE 32      // measured 32 / 40 / 52 with the product build's leaf test: C4 extend 21.5 / 22.2 / 26.7 ms per 64 spp, C5 13.5 / 14.05 / 15.8 per 32
#endif
// Wave-level scheduling of the walk.  The while-while loop (a node step for every walking lane, then primitive tests until the slowest
// lane has none left) kept half of the issue slots idle: most node steps leave a lane nothing to test, a few leave it a handful, and
// the wave pays for the longest list (measured, C4: lane utilisation 0.58, of which the primitive loop ran at ~0.25).  Now
// an iteration performs ONE kind of action, the one more lanes are waiting for - a node step (lanes whose pending primitive group is
// empty) or one primitive test (lanes with pending primitives) - and the others sit that iteration out; lanes gather on whichever side
// is the minority until it becomes the majority.
#ifndef APT_VOTE_TRI_WEIGHT
#define APT_VOTE_TRI_WEIGHT 2      // the vote is "primitive test if (lanes waiting for one) x weight >= lanes waiting for a node step"
#endif
#ifndef APT_VOTE_TRI_WEIGHT_SH
#define APT_VOTE_TRI_WEIGHT_SH APT_VOTE_TRI_WEIGHT      // the any-hit walk's weight (a primitive test that finds an occluder ends the lane's walk)
#endif
// after either action: a walking lane with nothing pending and no inner children left takes its next group from the stack, or is finished
APT_D void walk_settle(const TravStack& ts, int& sp, grp_t& ng, const grp_t& tg, int& state) {
    if (state == 1 && tg.y == 0u && !APT_GROUP_HAS_NODES(ng)) {
        if (sp == 0) state = 2; else ng = tpop(ts, sp);
    }
}
// Register budget of the walk kernels: left alone the allocator takes 86 VGPRs for the class-sorting closest-hit walk (five waves per SIMD);
// asked for seven waves it finds 70 without a spill (eight: 64 and a 20-byte spill).  Measured (product build, one lane, ms per 64 / 32 spp
// of C4 / C5): extend 18.73 -> 17.79 / 13.35 -> 13.04 at seven, 17.83 / 13.41 at eight, 18.03 / 13.23 at six.
#ifndef APT_WALK_WAVES
#define APT_WALK_WAVES 7
#endif
#if APT_WALK_WAVES > 0
#define APT_WALK_ATTR __attribute__((amdgpu_waves_per_eu(APT_WALK_WAVES, APT_WALK_WAVES)))
#else
#define APT_WALK_ATTR
#endif
template <int SORTED>
__global__ void __launch_bounds__(BLOCK) APT_WALK_ATTR k_extend_dyn(DevScene sc, Params p, Queues q, Counters* cnt, int cur_q, const uint32_t* n_src, LdsPlan plan) {
    const TravStack ts = make_walk_stack(plan);
    const int sq = (int)(blockIdx.x % (uint32_t)p.nq);
    const uint32_t n = n_src[sq * CNT_PAD];
    if (blockIdx.x / (uint32_t)p.nq == 0 && threadIdx.x == 0) {
        cnt->n_shadow[sq * CNT_PAD] = 0; cnt->n_active[cur_q ^ 1][sq * CNT_PAD] = 0;
        for (int w = 0; w < 8; w++) cnt->n_walk[w][sq * CNT_PAD] = 0;
        cnt->stats[sq][ST_EXTEND] += n;
        cnt->n_work[1][sq * CNT_PAD] = 0;                                          // this bounce's any-hit walk (k_shadow_dyn) starts at the head of its queue
    }
    uint32_t* work = &cnt->n_work[0][sq * CNT_PAD];
    lds_f* park = (lds_f*)(reinterpret_cast<float*>(s_dyn) + (size_t)plan.stack_depth * BLOCK * 2) + threadIdx.x;      // six floats per thread behind the stack ([component][thread]): the ray's throughput, id, meta, pdf while it walks
    const float* ro = q.ray_o[cur_q]; const float* rd = q.ray_d[cur_q];
    const uint32_t qbase = (uint32_t)sq * p.subcap;
    // per-lane ray state
    int state = 0;                              // 0 no ray, 1 walking, 2 finished (result not yet handed in)
    uint32_t io = 0;
    WalkRay r = make_walk_ray(sc.bvh, splat3(0.f), mk3(0.f, 0.f, 1.f));
    HitRec rec; rec.t = 1e7f; rec.prim = -1; rec.u = rec.v = 0.f;
    WalkStats ws; ws.nodes = ws.prims = 0;
    int sp = 0;
    grp_t ng = mk_grp(0u, 0u), tg = mk_grp(0u, 0u);       // node group / triangle group of the ray being walked (traverse.hpp)
    bool exhausted = false;                     // wave-uniform: the work counter has run past the queue
#ifdef APT_WALK_STATS
    uint32_t wd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        // ---- hand in finished rays (all lanes take part: the class appends are ballot-compacted)
        const bool fin = state == 2;
#if APT_FAST_LEAVES
        int fin_cls = -1;                                      // product build: the walk reports the winner's leaf slot; one lookup yields primitive and class
        if (fin) { const int info = walk_info(sc.bvh, rec.prim); rec.prim = info < 0 ? -1 : (info & 0x0fffffff); fin_cls = info < 0 ? -1 : (info >> 28); }
#endif
        if (!SORTED) {
            if (fin) { stq(q.hit_t, io, rec.t); stq(q.hit_prim, io, rec.prim); stq(q.hit_u, io, rec.u); stq(q.hit_v, io, rec.v); }
        } else if (__any(fin)) {
            // (as in k_extend: the rest of the record is requested first, the queue tails of all classes move with ONE atomic instruction)
            // (the rest of the path's record was fetched WITH the ray - six coalesced loads at the claim - and has waited in LDS behind the stack:
            // fetched here, by the finished lanes' scattered queue positions, it was six of the ~42 scattered accesses a ray makes)
            f3 st_thr = splat3(0.f); uint32_t st_id = 0, st_meta = 0; float st_pdf = 0.f;
            if (fin) { st_thr = mk3(park[0], park[BLOCK], park[2 * BLOCK]); st_id = __float_as_uint(park[3 * BLOCK]); st_meta = __float_as_uint(park[4 * BLOCK]); st_pdf = park[5 * BLOCK]; }
#if APT_FAST_LEAVES
            const int cls = (fin && rec.prim >= 0) ? fin_cls : -1;
#else
            const int cls = (fin && rec.prim >= 0) ? sc.prim_class[rec.prim] : -1;
#endif
            uint32_t my_rank = 0, cnt_vec = 0;
            for (int c = 0; c < q.n_classes; c++) {
                const unsigned long long m = __ballot(cls == c);
                if (cls == c) my_rank = rank_in(m);
                if ((int)lane_id() == c) cnt_vec = (uint32_t)__popcll(m);
            }
            uint32_t tail = 0;
            if ((int)lane_id() < q.n_classes && cnt_vec) tail = atomicAdd(&cnt->n_cls[lane_id()][sq * CNT_PAD], cnt_vec);
            const uint32_t cpos = (uint32_t)__shfl((int)tail, cls < 0 ? 0 : cls) + my_rank;
            if (cls >= 0) cq_store(q, (uint32_t)cls * p.cap + qbase + cpos, r.o, r.d, st_thr, st_id, st_meta, st_pdf, rec.t, rec.prim, rec.u, rec.v);
        }
        if (fin) state = 0;
        // ---- claim fresh rays for the idle lanes
        if (!exhausted) {
            const bool need = state == 0;
            const unsigned long long m = __ballot(need);
            uint32_t base = 0;
            if (lane_id() == 0 && m) base = atomicAdd(work, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);      // v_readlane: lane 0's value as a scalar (a shuffle would go through the LDS crossbar)
            const uint32_t pos = base + rank_in(m);
            if (need && pos < n) {
                io = (qbase + pos) << 2;
                r = make_walk_ray(sc.bvh, ld3q(ro, p.cap, io), ld3q(rd, p.cap, io));
                if (SORTED) {
                    const f3 t_ = ld3q(q.thr[cur_q], p.cap, io);
                    park[0] = t_.x; park[BLOCK] = t_.y; park[2 * BLOCK] = t_.z;
                    park[3 * BLOCK] = __uint_as_float(ldq(q.id[cur_q], io)); park[4 * BLOCK] = __uint_as_float(ldq(q.meta[cur_q], io)); park[5 * BLOCK] = ldq(q.pdf[cur_q], io);
                }
                rec.t = 1e7f; rec.prim = -1; rec.u = 0.f; rec.v = 0.f;
                sp = 0; ng = APT_ROOT_GROUP; state = 1;
            }
            if (base + (uint32_t)__popcll(m) >= n) exhausted = true;
#ifdef APT_WALK_STATS
            wd[5]++; wd[6] += (uint32_t)__popcll(__ballot(need && pos < n));
#endif
        }
        if (!__any(state == 1)) break;
        // ---- walk: the while-while loop of traverse<false>, left as soon as too few lanes still hold a ray
        const uint32_t min_active = exhausted ? 1u : (uint32_t)APT_DYN_MIN_ACTIVE;
        // One action per iteration for the whole wave, chosen by vote (see walk_settle): a node step for the lanes without pending primitives,
        // or one primitive test for the lanes with some.  Per ray nothing changes - same nodes, same primitives, same order.
        do {
            const bool want_t = state == 1 && tg.y != 0u, want_n = state == 1 && tg.y == 0u;
#ifdef APT_WALK_STATS
            { const uint32_t a_ = (uint32_t)__popcll(__ballot(want_t)), b_ = (uint32_t)__popcll(__ballot(want_n)); wd[4] += a_ + b_; if (a_ * APT_VOTE_TRI_WEIGHT >= b_) { wd[2]++; wd[3] += a_; } else { wd[0]++; wd[1] += b_; } }
#endif
            if (__popcll(__ballot(want_t)) * APT_VOTE_TRI_WEIGHT >= __popcll(__ballot(want_n))) { if (want_t) tri_one<false>(sc.bvh, tg, r, rec, ws); }
            else if (want_n) group_step(sc.bvh, ts, sp, ng, tg, r, rec.t, ws);
            walk_settle(ts, sp, ng, tg, state);
        } while ((uint32_t)__popcll(__ballot(state == 1)) >= min_active);
    }
#ifdef APT_WALK_STATS
    flush_stat(ws.nodes, &cnt->stats[sq][10]); flush_stat(ws.prims, &cnt->stats[sq][11]);
    if (lane_id() == 0) for (int k = 0; k < 7; k++) atomicAdd(&cnt->wdbg[0][k], (unsigned long long)wd[k]);
#endif
}

// ------------------------------------------------------------------- shadow
template <int MODE>
__global__ void __launch_bounds__(TRACE_NT(MODE), (MODE == 2 ? APT_TILE_WAVES : 1)) k_shadow(DevScene sc, Params p, Queues q, Counters* cnt, LdsPlan plan) {
    __shared__ float s_sweep[MODE == 1 ? APT_SWEEP_LDS_FLOATS(BLOCK) : 1];
    const SubLoop sl = sub_loop(p.nq, TRACE_NT(MODE));
    const uint32_t n = min(cnt->n_shadow[sl.q * CNT_PAD], q.sh_subcap);
    if (sl.first == 0 && threadIdx.x == 0) {
        cnt->stats[sl.q][ST_SHADOW_TRACED] += n;
        for (int c = 0; c < q.n_classes; c++) cnt->n_cls[c][sl.q * CNT_PAD] = 0;      // every shade of this bounce is done
    }
    const uint32_t qbase = (uint32_t)sl.q * q.sh_subcap, sc_ = q.sh_cap;
    uint32_t t_lit = 0;
    for (uint32_t base = sl.first; base < n; base += sl.stride) {
        const uint32_t pos = base + threadIdx.x;
        const bool valid = pos < n;
        const uint32_t idx = qbase + (valid ? pos : n - 1);
        const uint32_t io = idx << 2;
        const f3 o = ld3q(q.sh_o, sc_, io);
        const f3 d = ld3q(q.sh_d, sc_, io);
        const float dist = ldq(q.sh_tmax, io);
        HitRec rec; rec.t = (dist > 0.0f) ? dist - 1e-4f : 1e7f; rec.prim = -1; rec.u = rec.v = 0.f;
        const bool occluded = (MODE == 0) ? traverse<true>(sc.bvh, make_stack(plan), o, d, rec)
                            : (MODE == 1) ? sweep_wg<true, BLOCK>(sc.sweep, o, d, rec, valid, s_sweep)
                                          : sweep_tile<true, APT_TILE_NT>(sc.sweep, o, d, rec, valid, reinterpret_cast<float*>(s_dyn));
        if (valid) {
            f3 c = ld3q(q.sh_c, sc_, io);
            // Upstream an occluded light sample still enters the sum as 0 * throughput; with a non-finite
            // throughput (pdf == 0 upstream, quirk A.3 #11) that is NaN, so the sample component is zeroed at
            // the end.  c carries the throughput factor: c * 0 reproduces exactly that, and is 0 for finite c.
            const bool weird = !(isfinite(c.x) && isfinite(c.y) && isfinite(c.z));
            if (occluded && weird) c = c * 0.f;
            if (!occluded || weird) {
                add_radiance(q.L, p.cap, ldq(q.sh_id, io), c, APT_EXCLUSIVE_L(p));      // sh_id: byte offset of the path's radiance slot
            }
            if (!occluded) t_lit++;
        }
    }
    flush_stat(t_lit, &cnt->stats[sl.q][ST_LIT]);
}

// ------------------------------------------------------- shadow, BVH walk with dynamic ray fetch
// Any-hit twin of k_extend_dyn: a lane leaves the walk at its first occluder or when its stack runs empty, adds its contribution if
// unoccluded, and claims the next shadow ray as soon as the wave runs low on walking lanes.
__global__ void __launch_bounds__(BLOCK) APT_WALK_ATTR k_shadow_dyn(DevScene sc, Params p, Queues q, Counters* cnt, LdsPlan plan) {
    const TravStack ts = make_walk_stack(plan);
    const int sq = (int)(blockIdx.x % (uint32_t)p.nq);
    const uint32_t n = min(cnt->n_shadow[sq * CNT_PAD], q.sh_subcap);
    if (blockIdx.x / (uint32_t)p.nq == 0 && threadIdx.x == 0) {
        cnt->stats[sq][ST_SHADOW_TRACED] += n;
        for (int c = 0; c < q.n_classes; c++) cnt->n_cls[c][sq * CNT_PAD] = 0;      // every shade of this bounce is done
        cnt->n_work[0][sq * CNT_PAD] = 0;                                          // the next bounce's closest-hit walk starts at the head of its queue (a host-side fill per bounce was a launch of its own: 15 us)
    }
    uint32_t* work = &cnt->n_work[1][sq * CNT_PAD];
    lds_f* park = (lds_f*)(reinterpret_cast<float*>(s_dyn) + (size_t)plan.stack_depth * BLOCK * 2) + threadIdx.x;      // four floats per thread behind the stack: the light sample's contribution and radiance slot while its ray walks
    const uint32_t qbase = (uint32_t)sq * q.sh_subcap, sc_ = q.sh_cap;
    int state = 0;                              // 0 no ray, 1 walking, 2 finished
    bool occluded = false;
    uint32_t io = 0;
    WalkRay r = make_walk_ray(sc.bvh, splat3(0.f), mk3(0.f, 0.f, 1.f));
    HitRec rec; rec.t = 0.f; rec.prim = -1; rec.u = rec.v = 0.f;      // rec.t = the search limit (distance to the light - 1e-4)
    WalkStats ws; ws.nodes = ws.prims = 0;
    int sp = 0;
    grp_t ng = mk_grp(0u, 0u), tg = mk_grp(0u, 0u);
    bool exhausted = false;
    uint32_t t_lit = 0;
#ifdef APT_WALK_STATS
    uint32_t wd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        if (state == 2) {
            f3 c = mk3(park[0], park[BLOCK], park[2 * BLOCK]);      // (contribution and slot were fetched with the ray - coalesced - and have waited in LDS: k_extend_dyn)
            const bool weird = !(isfinite(c.x) && isfinite(c.y) && isfinite(c.z));       // see k_shadow
            if (occluded && weird) c = c * 0.f;
            if (!occluded || weird) {
                add_radiance(q.L, p.cap, __float_as_uint(park[3 * BLOCK]), c, APT_EXCLUSIVE_L(p));
            }
            if (!occluded) t_lit++;
            state = 0;
        }
        if (!exhausted) {
            const bool need = state == 0;
            const unsigned long long m = __ballot(need);
            uint32_t base = 0;
            if (lane_id() == 0 && m) base = atomicAdd(work, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);      // v_readlane: lane 0's value as a scalar (a shuffle would go through the LDS crossbar)
            const uint32_t pos = base + rank_in(m);
            if (need && pos < n) {
                io = (qbase + pos) << 2;
                r = make_walk_ray(sc.bvh, ld3q(q.sh_o, sc_, io), ld3q(q.sh_d, sc_, io));
                const float dist = ldq(q.sh_tmax, io);
                { const f3 c_ = ld3q(q.sh_c, sc_, io); park[0] = c_.x; park[BLOCK] = c_.y; park[2 * BLOCK] = c_.z; park[3 * BLOCK] = __uint_as_float(ldq(q.sh_id, io)); }
                rec.t = (dist > 0.0f) ? dist - 1e-4f : 1e7f;
                occluded = false; sp = 0; ng = APT_ROOT_GROUP; state = 1;
            }
            if (base + (uint32_t)__popcll(m) >= n) exhausted = true;
#ifdef APT_WALK_STATS
            wd[5]++; wd[6] += (uint32_t)__popcll(__ballot(need && pos < n));
#endif
        }
        if (!__any(state == 1)) break;
        const uint32_t min_active = exhausted ? 1u : (uint32_t)APT_DYN_MIN_ACTIVE_SH;
        do {
            const bool want_t = state == 1 && tg.y != 0u, want_n = state == 1 && tg.y == 0u;
#ifdef APT_WALK_STATS
            { const uint32_t a_ = (uint32_t)__popcll(__ballot(want_t)), b_ = (uint32_t)__popcll(__ballot(want_n)); wd[4] += a_ + b_; if (a_ * APT_VOTE_TRI_WEIGHT_SH >= b_) { wd[2]++; wd[3] += a_; } else { wd[0]++; wd[1] += b_; } }
#endif
            if (__popcll(__ballot(want_t)) * APT_VOTE_TRI_WEIGHT_SH >= __popcll(__ballot(want_n))) {
                if (want_t && tri_one<true>(sc.bvh, tg, r, rec, ws)) { occluded = true; sp = 0; ng.y = 0u; tg.y = 0u; }      // first occluder ends the walk
            } else if (want_n) group_step(sc.bvh, ts, sp, ng, tg, r, rec.t, ws);
            walk_settle(ts, sp, ng, tg, state);
        } while ((uint32_t)__popcll(__ballot(state == 1)) >= min_active);
    }
    flush_stat(t_lit, &cnt->stats[sq][ST_LIT]);
#ifdef APT_WALK_STATS
    flush_stat(ws.nodes, &cnt->stats[sq][12]); flush_stat(ws.prims, &cnt->stats[sq][13]);
    if (lane_id() == 0) for (int k = 0; k < 7; k++) atomicAdd(&cnt->wdbg[1][k], (unsigned long long)wd[k]);
#endif
}

// ------------------------------------------------------- flat sweep stages (fast build, small scenes)
// Two queue entries per lane (traverse.hpp "Flat sweep"): lane k of a 256-thread workgroup owns entries 2k and 2k + 1 of a 512-entry
// tile, every record component arrives as one 8-byte load, and what the lane writes back is 8 bytes per component too.
#if APT_FAST
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
template <typename T2> APT_D T2 ld2q(const void* base, uint32_t off) { return *reinterpret_cast<const T2*>(reinterpret_cast<const char*>(base) + off); }
template <typename T2> APT_D void st2q(void* base, uint32_t off, T2 v) { *reinterpret_cast<T2*>(reinterpret_cast<char*>(base) + off) = v; }
#define FLAT_NT (2 * BLOCK)

// Fix-up lists.  Two kinds of rays need the reference's own arithmetic (traverse.hpp flat_closest2): they are a handful per million, but
// the code that serves them - prim_test(), the reference-order sweep() - inlined into the stage kernels set their register allocation
// (k_extend_flat 79 VGPRs, k_shadow_flat 85: 5-6 waves per SIMD).  So the kernels come in three variants: VAR 1, the hot one, resolves
// every other ray and appends the entries it cannot settle to a per-sub-queue list (52 / 44 VGPRs); VAR 2 runs right after it on the same
// stream with a small grid, takes its entries from that list - one per lane - and does the full work for them, including the class
// appends of the sorted pipeline; VAR 0 is the self-contained kernel for explicit rays (apt_intersect), where no lists exist.
APT_D void fix_append(bool f0, bool f1, uint32_t pos, uint32_t* counter, uint32_t* list, uint32_t qbase) {
    if (!__any(f0 || f1)) return;
    const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
    uint32_t tail = 0;
    if (lane_id() == 0) tail = atomicAdd(counter, (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1));
    const uint32_t at = (qbase + (uint32_t)__builtin_amdgcn_readlane((int)tail, 0) + rank_in(m0) + rank_in(m1)) << 2;
    if (f0) stq(list, at, pos);
    if (f1) stq(list, at + (f0 ? 4u : 0u), pos + 1u);
}
template <int SORTED, int VAR>
APT_D void extend_flat_body(const DevScene& sc, const Params& p, const Queues& q, Counters* cnt, int cur, const uint32_t* n_src) {
    const SubLoop sl = sub_loop(p.nq, VAR == 2 ? BLOCK : FLAT_NT);
    const uint32_t n = (VAR == 2) ? cnt->n_fix_ext[cur][sl.q * CNT_PAD] : n_src[sl.q * CNT_PAD];
    if (VAR != 2 && cnt && sl.first == 0 && threadIdx.x == 0) {
        cnt->n_shadow[sl.q * CNT_PAD] = 0; cnt->n_active[cur ^ 1][sl.q * CNT_PAD] = 0;
        for (int w = 0; w < 8; w++) cnt->n_walk[w][sl.q * CNT_PAD] = 0;
        cnt->n_fix_ext[cur ^ 1][sl.q * CNT_PAD] = 0; cnt->n_fix_sh[cur][sl.q * CNT_PAD] = 0;   // (both lists were consumed by fix-up launches that have finished)
        cnt->stats[sl.q][ST_EXTEND] += n;
    }
    const float* ro = q.ray_o[cur]; const float* rd = q.ray_d[cur];
    const uint32_t qbase = (uint32_t)sl.q * p.subcap, cs = p.cap * 4u;             // component stride in bytes
    for (uint32_t base = sl.first; base < n; base += sl.stride) {
        uint32_t pos; bool v0, v1; uint32_t io; bool odd = false;
        if (VAR == 2) {                                                            // one listed entry per lane, as "entry 0" of its pair
            const uint32_t li = base + threadIdx.x;
            v0 = li < n; v1 = false;
            pos = ldq(q.fix_ext, (qbase + (v0 ? li : n - 1u)) << 2);
            odd = (pos & 1u) != 0u; io = (qbase + (pos & ~1u)) << 2;
        } else {
            pos = base + 2u * threadIdx.x;
            v0 = pos < n; v1 = pos + 1u < n;
            io = (qbase + (v0 ? pos : ((n - 1u) & ~1u))) << 2;                      // idle lanes re-read the last pair (never written back)
        }
        v2f ox = ld2q<v2f>(ro, io), oy = ld2q<v2f>(ro, io + cs), oz = ld2q<v2f>(ro, io + 2u * cs);
        v2f dx = ld2q<v2f>(rd, io), dy = ld2q<v2f>(rd, io + cs), dz = ld2q<v2f>(rd, io + 2u * cs);
        if (VAR == 2 && odd) { ox = mk2(ox.y, ox.x); oy = mk2(oy.y, oy.x); oz = mk2(oz.y, oz.x); dx = mk2(dx.y, dx.x); dy = mk2(dy.y, dy.x); dz = mk2(dz.y, dz.x); }
        const uint32_t io0 = (VAR == 2) ? (qbase + pos) << 2 : io;                  // byte offset of entry 0's own slot
        const f3 o0 = mk3(ox.x, oy.x, oz.x), d0 = mk3(dx.x, dy.x, dz.x), o1 = mk3(ox.y, oy.y, oz.y), d1 = mk3(dx.y, dy.y, dz.y);
        HitRec r0, r1; r0.t = r1.t = 1e7f; r0.prim = r1.prim = -1; r0.u = r0.v = r1.u = r1.v = 0.f;
        int c0, c1; bool sp0, sp1;
        flat_closest2<VAR == 1>(sc.flat, sc.sweep, sc.prim_class, o0, d0, o1, d1, r0, r1, c0, c1, sp0, sp1);
        if (VAR == 1) { sp0 = sp0 && v0; sp1 = sp1 && v1; fix_append(sp0, sp1, pos, &cnt->n_fix_ext[cur][sl.q * CNT_PAD], q.fix_ext, qbase); }
        if (!SORTED) {
            // barycentrics only travel when somebody reads them: vertex normals, textures, or the unit-test entry (apt_intersect: cnt == nullptr)
            const bool need_uv = sc.has_vn || sc.tex_i != nullptr || cnt == nullptr;
            if (v1) {
                st2q<v2f>(q.hit_t, io, mk2(r0.t, r1.t)); v2i pr; pr.x = r0.prim; pr.y = r1.prim; st2q<v2i>(q.hit_prim, io, pr);
                if (need_uv) { st2q<v2f>(q.hit_u, io, mk2(r0.u, r1.u)); st2q<v2f>(q.hit_v, io, mk2(r0.v, r1.v)); }
            } else if (v0) { stq(q.hit_t, io0, r0.t); stq(q.hit_prim, io0, r0.prim); if (need_uv) { stq(q.hit_u, io0, r0.u); stq(q.hit_v, io0, r0.v); } }
            // (VAR 1 writes a provisional record for a deferred entry; the fix-up launch overwrites it before anybody reads it)
        } else {
            // sort by material class (see k_extend): the tails of all class queues move with ONE atomic instruction per tile row
            v2f tx = ld2q<v2f>(q.thr[cur], io), ty = ld2q<v2f>(q.thr[cur], io + cs), tz = ld2q<v2f>(q.thr[cur], io + 2u * cs);
            v2u pid = ld2q<v2u>(q.id[cur], io), pmeta = ld2q<v2u>(q.meta[cur], io); v2f ppdf = ld2q<v2f>(q.pdf[cur], io);
            if (VAR == 2 && odd) { tx = mk2(tx.y, tx.x); ty = mk2(ty.y, ty.x); tz = mk2(tz.y, tz.x); ppdf = mk2(ppdf.y, ppdf.x); v2u t_; t_.x = pid.y; t_.y = pid.x; pid = t_; t_.x = pmeta.y; t_.y = pmeta.x; pmeta = t_; }
            // (a deferred entry joins its class queue in the fix-up launch)
            const int cls0 = (v0 && !sp0 && r0.prim >= 0) ? c0 : -1, cls1 = (v1 && !sp1 && r1.prim >= 0) ? c1 : -1;
            uint32_t rank0 = 0, rank1 = 0, cnt_vec = 0;
            for (int c = 0; c < q.n_classes; c++) {
                const unsigned long long m0 = __ballot(cls0 == c), m1 = __ballot(cls1 == c);
                // queue order: the wave's first entries, then its second ones - each record store then writes consecutive slots (C3's extend 7.2
                // against 8.4 ms per 128 spp with a lane's two entries kept neighbours) and nothing downstream cares
                const uint32_t n0 = (uint32_t)__popcll(m0);
                if (cls0 == c) rank0 = rank_in(m0);
                if (cls1 == c) rank1 = n0 + rank_in(m1);
                if ((int)lane_id() == c) cnt_vec = (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
            }
            uint32_t tail = 0;
            if ((int)lane_id() < q.n_classes && cnt_vec) tail = atomicAdd(&cnt->n_cls[lane_id()][sl.q * CNT_PAD], cnt_vec);
            const uint32_t so0 = (qbase + (uint32_t)__shfl((int)tail, cls0 < 0 ? 0 : cls0) + rank0) << 2;
            const uint32_t so1 = (qbase + (uint32_t)__shfl((int)tail, cls1 < 0 ? 0 : cls1) + rank1) << 2;
            if (cls0 >= 0) cq_store(q, (uint32_t)cls0 * p.cap + (so0 >> 2), o0, d0, mk3(tx.x, ty.x, tz.x), pid.x, pmeta.x, ppdf.x, r0.t, r0.prim, r0.u, r0.v);
            if (cls1 >= 0) cq_store(q, (uint32_t)cls1 * p.cap + (so1 >> 2), o1, d1, mk3(tx.y, ty.y, tz.y), pid.y, pmeta.y, ppdf.y, r1.t, r1.prim, r1.u, r1.v);
        }
    }
}

template <int SORTED, int VAR>
__global__ void __launch_bounds__(BLOCK) k_extend_flat(DevScene sc, Params p, Queues q, Counters* cnt, int cur, const uint32_t* n_src, LdsPlan plan) {
    extend_flat_body<SORTED, VAR>(sc, p, q, cnt, cur, n_src);
}

// par: parity of the shadow fix-up list this launch appends to (VAR 1: the bounce's queue parity, Params::fix_par) or consumes (VAR 2)
template <int VAR>
APT_D void shadow_flat_body(const DevScene& sc, const Params& p, const Queues& q, Counters* cnt, int par) {
    // VAR 1 / 2: hot variant and the fix-up pass (see k_extend_flat); an entry whose ray needs the reference-order sweep is listed by
    // VAR 1 - untouched: no radiance, no statistics - and done in full by VAR 2.
    // VAR 3: the fix-up pass of a render whose shade kernel traces its rays itself (shade_stage.hpp k_shade_traced): the shadow queue then holds
    // nothing but the deferred entries, n_fix_sh[par] of them, and every one is done in full, one per lane, like VAR 2's.
    const SubLoop sl = sub_loop(p.nq, VAR >= 2 ? BLOCK : FLAT_NT);
    const uint32_t n = (VAR == 2) ? cnt->n_fix_sh[par][sl.q * CNT_PAD] : ((VAR == 3) ? min(cnt->n_fix_sh[par][sl.q * CNT_PAD], q.sh_subcap) : min(cnt->n_shadow[sl.q * CNT_PAD], q.sh_subcap));
    if (VAR < 2 && sl.first == 0 && threadIdx.x == 0) {
        if (!p.nee_vm) cnt->stats[sl.q][ST_SHADOW_TRACED] += n;
        for (int c = 0; c < q.n_classes; c++) cnt->n_cls[c][sl.q * CNT_PAD] = 0;      // every shade of this bounce is done
    }
    const uint32_t qbase = (uint32_t)sl.q * q.sh_subcap, cs = q.sh_cap * 4u;
    uint32_t* fix_counter = &cnt->n_fix_sh[par][sl.q * CNT_PAD];
    uint32_t t_lit = 0, t_traced = 0;
    if (p.nee_vm) {
        // Light samples by vertex (S > 1; `n` counts VERTICES): a lane owns one vertex and walks its sample planes two at a time - the two
        // rays of a packed test then share their origin.  The samples are summed in sample order, as upstream sums direct_int, and the
        // vertex's radiance slot takes ONE read-modify-write, which no other lane of the launch touches.  (tmax < 0: a sample the shade
        // kernel found not worth tracing.)  An occluded sample still enters the sum as 0 * contribution - NaN for a non-finite one (see k_shadow).
        const SubLoop sv = sub_loop(p.nq, BLOCK);
        for (uint32_t base = sv.first; base < n; base += sv.stride) {
            const uint32_t li = base + threadIdx.x;
            const bool valid = li < n;
            const uint32_t pos = (VAR == 2) ? ldq(q.fix_sh, (qbase + (valid ? li : n - 1u)) << 2) : (valid ? li : n - 1u);
            const uint32_t io = (qbase + pos) << 2;
            const uint32_t slot = ldq(q.sh_id, io);
            const f3 o = ld3q(q.sh_o, q.sh_cap, io);                              // (one origin per vertex, with its first entry)
            f3 sum = splat3(0.f); bool any = false, defer = false;
            uint32_t v_traced = 0, v_lit = 0;
            for (int smp = 0; smp < p.S; smp += 2) {
                const bool two = smp + 1 < p.S;
                const uint32_t ia = io + ((uint32_t)smp * p.subcap << 2), ib = io + ((uint32_t)(two ? smp + 1 : smp) * p.subcap << 2);
                const float ta = ldq(q.sh_tmax, ia), tb = ldq(q.sh_tmax, ib);
                const bool la = valid && !(ta < 0.0f), lb = valid && two && !(tb < 0.0f);
                const f3 da = ld3q(q.sh_d, q.sh_cap, ia), db = ld3q(q.sh_d, q.sh_cap, ib);
                bool oa, ob, spa, spb;
                flat_any2<VAR == 1>(sc.flat, sc.sweep, o, da, o, db, la ? ((ta > 0.0f) ? ta - 1e-4f : 1e7f) : -1.0f, lb ? ((tb > 0.0f) ? tb - 1e-4f : 1e7f) : -1.0f, oa, ob, spa, spb);
                if (VAR == 1) defer = defer || (la && spa) || (lb && spb);
                f3 ca = ld3q(q.sh_c, q.sh_cap, ia), cb = ld3q(q.sh_c, q.sh_cap, ib);
                const bool wa = !(isfinite(ca.x) && isfinite(ca.y) && isfinite(ca.z)), wb = !(isfinite(cb.x) && isfinite(cb.y) && isfinite(cb.z));
                if (la && (!oa || wa)) { if (oa) ca = ca * 0.f; sum = any ? sum + ca : ca; any = true; }
                if (lb && (!ob || wb)) { if (ob) cb = cb * 0.f; sum = any ? sum + cb : cb; any = true; }
                v_traced += (la ? 1u : 0u) + (lb ? 1u : 0u); v_lit += (la && !oa ? 1u : 0u) + (lb && !ob ? 1u : 0u);
            }
            if (VAR == 1) fix_append(defer, false, pos, fix_counter, q.fix_sh, qbase);      // the whole vertex goes to the fix-up launch
            if (!defer) {
                if (any) add_radiance(q.L, p.cap, slot & ~3u, sum, true);
                t_traced += v_traced; t_lit += v_lit;
            }
        }
        flush_stat(t_lit, &cnt->stats[sl.q][ST_LIT]);
        flush_stat(t_traced, &cnt->stats[sl.q][ST_SHADOW_TRACED]);
        return;
    }
    for (uint32_t base = sl.first; base < n; base += sl.stride) {
        uint32_t pos; bool v0, v1; uint32_t io; bool odd = false;
        if (VAR >= 2) {
            const uint32_t li = base + threadIdx.x;
            v0 = li < n; v1 = false;
            pos = (VAR == 2) ? ldq(q.fix_sh, (qbase + (v0 ? li : n - 1u)) << 2) : (v0 ? li : n - 1u);
            odd = (pos & 1u) != 0u; io = (qbase + (pos & ~1u)) << 2;
        } else {
            pos = base + 2u * threadIdx.x;
            v0 = pos < n; v1 = pos + 1u < n;
            io = (qbase + (v0 ? pos : ((n - 1u) & ~1u))) << 2;
        }
        v2f ox = ld2q<v2f>(q.sh_o, io), oy = ld2q<v2f>(q.sh_o, io + cs), oz = ld2q<v2f>(q.sh_o, io + 2u * cs);
        v2f dx = ld2q<v2f>(q.sh_d, io), dy = ld2q<v2f>(q.sh_d, io + cs), dz = ld2q<v2f>(q.sh_d, io + 2u * cs);
        v2f dist = ld2q<v2f>(q.sh_tmax, io);
        // the radiance slots are requested with the rays, so that after the sweep ONE round trip fetches contribution and radiance of both entries
        v2u slot = ld2q<v2u>(q.sh_id, io);
        if (VAR >= 2 && odd) {
            ox = mk2(ox.y, ox.x); oy = mk2(oy.y, oy.x); oz = mk2(oz.y, oz.x); dx = mk2(dx.y, dx.x); dy = mk2(dy.y, dy.x); dz = mk2(dz.y, dz.x);
            dist = mk2(dist.y, dist.x); v2u t_; t_.x = slot.y; t_.y = slot.x; slot = t_;
        }
        bool occ0, occ1, sp0, sp1;
        flat_any2<VAR == 1>(sc.flat, sc.sweep, mk3(ox.x, oy.x, oz.x), mk3(dx.x, dy.x, dz.x), mk3(ox.y, oy.y, oz.y), mk3(dx.y, dy.y, dz.y),
                            (dist.x > 0.0f) ? dist.x - 1e-4f : 1e7f, (dist.y > 0.0f) ? dist.y - 1e-4f : 1e7f, occ0, occ1, sp0, sp1);
        if (VAR == 1) {
            sp0 = sp0 && v0; sp1 = sp1 && v1;
            fix_append(sp0, sp1, pos, fix_counter, q.fix_sh, qbase);
            v0 = v0 && !sp0; v1 = v1 && !sp1;                                    // a listed entry is left alone here
        }
        v2f cx = ld2q<v2f>(q.sh_c, io), cy = ld2q<v2f>(q.sh_c, io + cs), cz = ld2q<v2f>(q.sh_c, io + 2u * cs);
        if (VAR >= 2 && odd) { cx = mk2(cx.y, cx.x); cy = mk2(cy.y, cy.x); cz = mk2(cz.y, cz.x); }
        const bool excl = APT_EXCLUSIVE_L(p);
#ifdef APT_PROBE_NO_L      // measurement only (tools/build_variant.sh nol -DAPT_PROBE_NO_L=1): how much of the stage is the radiance read-modify-write
        if (cx.x == 123.456f) stL(q.L, p.cap, slot.x, ldL(q.L, p.cap, slot.x));
        t_lit += (v0 && !occ0 ? 1u : 0u) + (v1 && !occ1 ? 1u : 0u);
        continue;
#endif
        // no two entries of the launch share a slot (excl): plain read-modify-writes of one 16-byte slot each; the radiance of both entries is
        // requested together with the contributions (whether it will be written is only known once they arrive)
        f3 a0 = splat3(0.f), a1 = splat3(0.f);
        if (excl && v0 && !occ0) a0 = ldL(q.L, p.cap, slot.x);
        if (excl && v1 && !occ1) a1 = ldL(q.L, p.cap, slot.y);
        // see k_shadow: an occluded sample still enters the sum as 0 * contribution, which is NaN for a non-finite contribution
        f3 c0 = mk3(cx.x, cy.x, cz.x), c1 = mk3(cx.y, cy.y, cz.y);
        const bool weird0 = !(isfinite(c0.x) && isfinite(c0.y) && isfinite(c0.z)), weird1 = !(isfinite(c1.x) && isfinite(c1.y) && isfinite(c1.z));
        if (excl) {
            if (v0 && !occ0) stL(q.L, p.cap, slot.x, mk3(a0.x + c0.x, a0.y + c0.y, a0.z + c0.z));
            else if (v0 && weird0) add_radiance(q.L, p.cap, slot.x, c0 * 0.f, true);          // rare: poisons the slot
            if (v1 && !occ1) stL(q.L, p.cap, slot.y, mk3(a1.x + c1.x, a1.y + c1.y, a1.z + c1.z));
            else if (v1 && weird1) add_radiance(q.L, p.cap, slot.y, c1 * 0.f, true);
        } else {
            if (v0 && (!occ0 || weird0)) add_radiance(q.L, p.cap, slot.x, (occ0 && weird0) ? c0 * 0.f : c0, false);
            if (v1 && (!occ1 || weird1)) add_radiance(q.L, p.cap, slot.y, (occ1 && weird1) ? c1 * 0.f : c1, false);
        }
        t_lit += (v0 && !occ0 ? 1u : 0u) + (v1 && !occ1 ? 1u : 0u);
    }
    flush_stat(t_lit, &cnt->stats[sl.q][ST_LIT]);
}
template <int VAR>
__global__ void __launch_bounds__(BLOCK) k_shadow_flat(DevScene sc, Params p, Queues q, Counters* cnt, LdsPlan plan) {
    shadow_flat_body<VAR>(sc, p, q, cnt, p.fix_par);
}
// The fix-up launch of a bounce, right after its hot extend kernel: the entries that kernel listed, and the shadow entries the PREVIOUS
// bounce's hot shadow kernel listed (one extra launch per bounce instead of two: every launch boundary is a pipeline drain; a last one
// after the final bounce, with an empty extend list, serves that bounce's shadow list).
template <int SORTED>
__global__ void __launch_bounds__(BLOCK) k_fix_flat(DevScene sc, Params p, Queues q, Counters* cnt, int cur, const uint32_t* n_src, LdsPlan plan) {
    extend_flat_body<SORTED, 2>(sc, p, q, cnt, cur, n_src);
    if (p.S > 0 && !p.volumetric_flat) { if (p.fused) shadow_flat_body<3>(sc, p, q, cnt, cur ^ 1); else shadow_flat_body<2>(sc, p, q, cnt, cur ^ 1); }
}

__global__ void __launch_bounds__(BLOCK) k_occluded_flat(DevScene sc, uint32_t n, const float* o_, const float* d_, const float* tmax, int* occ, LdsPlan plan) {
    for (uint32_t base = blockIdx.x * FLAT_NT; base < n; base += gridDim.x * FLAT_NT) {
        const uint32_t pos = base + 2u * threadIdx.x;
        const bool v0 = pos < n, v1 = pos + 1u < n;
        const uint32_t i0 = v0 ? pos : n - 1, i1 = v1 ? pos + 1u : n - 1;
        const f3 o0 = mk3(o_[i0], o_[n + i0], o_[2 * n + i0]), d0 = mk3(d_[i0], d_[n + i0], d_[2 * n + i0]);
        const f3 o1 = mk3(o_[i1], o_[n + i1], o_[2 * n + i1]), d1 = mk3(d_[i1], d_[n + i1], d_[2 * n + i1]);
        bool a, b;
        flat_any2(sc.flat, sc.sweep, o0, d0, o1, d1, (tmax[i0] > 0.0f) ? tmax[i0] - 1e-4f : 1e7f, (tmax[i1] > 0.0f) ? tmax[i1] - 1e-4f : 1e7f, a, b);
        if (v0) occ[i0] = a ? 1 : 0;
        if (v1) occ[i1] = b ? 1 : 0;
    }
}
#endif

// ----------------------------------------------------------------- finalize
// one thread per owned pixel: samples summed in sample order -> bit-reproducible, no atomics
__global__ void __launch_bounds__(BLOCK) k_finalize(Params p, Queues q, float* accum) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t lp = blockIdx.x * BLOCK + threadIdx.x; lp < (uint32_t)p.npix; lp += stride) {
        float r = accum[3 * lp], g = accum[3 * lp + 1], b = accum[3 * lp + 2];
        for (int s = 0; s < p.spp_batch; s++) {
            const uint32_t lo_ = ((uint32_t)s * (uint32_t)p.npix + lp) << 2;
            f3 c_ = ldL(q.L, p.cap, lo_);
            for (int pl = 1; pl < p.l_planes; pl++) c_ = c_ + ldL(q.L, p.cap, lo_ | (uint32_t)pl);      // planes in light-sample order
            float cr = c_.x, cg = c_.y, cb = c_.z;
            r += isnan(cr) ? 0.f : cr; g += isnan(cg) ? 0.f : cg; b += isnan(cb) ? 0.f : cb;
        }
        accum[3 * lp] = r; accum[3 * lp + 1] = g; accum[3 * lp + 2] = b;
    }
}

__global__ void k_divide(const float* accum, float* out, uint32_t n, float cnt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = accum[i] / cnt;        // pixels = color / cnt (vanilla_renderer.py:120)
}
