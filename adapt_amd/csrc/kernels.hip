// kernels.hip — the wavefront path tracer: stage kernels + host driver + C-ABI (include/adapt_mi.h).
//
// Hot path being replaced: AdaPT's megakernel `Renderer.render`
// (renderer/vanilla_renderer.py:32-120, one launch per spp, one thread per pixel carrying a
// whole path).  Here the same path-space computation is decomposed into stages that each
// stream a dense SoA queue in HBM:
//
//   generate : camera ray per (pixel, sample) slot                       tracer_base.py:136-157
//   extend   : closest hit for every queued ray                          tracer_base.py:168-237 / path_tracer.py:338-394
//   shade    : emission + MIS, RR, NEE sampling -> shadow queue,         vanilla_renderer.py:44-117
//              BSDF sampling -> next ray queue (ballot-compacted)
//   shadow   : any-hit for every shadow ray, unoccluded ones add         tracer_base.py:239-278 / path_tracer.py:396-422
//              their contribution to the owning path's radiance
//   finalize : per pixel, sum the batch's samples in sample order,       vanilla_renderer.py:119-120
//              NaN -> 0, accumulate into the float3 framebuffer
//
// A slot id = sample_in_batch * n_local_pixels + local_pixel identifies a path for its whole
// life: the RNG is keyed by (global pixel, sample counter) and only a draw index is carried,
// and the radiance accumulator L[] is indexed by id, so results do not depend on where a
// path sits in a queue.  See DESIGN.md for the HBM layout and the bytes each stage moves.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/adapt_mi.h"
#include "bvh_build.hpp"
#include "rng.hpp"
#include "shading.hpp"
#include "traverse.hpp"
#include "vec.hpp"

#define APT_EXPORT extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(APT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

// ------------------------------------------------------------- device views
struct DevScene {
    DevBvh bvh;
    const float* normals;     // n_prims*3
    const float* vnormals;    // n_prims*9
    const float* precom;      // n_prims*9
    const int* prim_obj;      // n_prims
    const int* obj_info;      // n_objects*3
    const int* emitter_id;    // n_objects
    const DevBxdf* bxdf;      // n_objects
    const DevSrc* src;        // n_sources
    int n_prims, n_objects, n_sources, has_vn;
    float world_ior;
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
    uint32_t cap;
};

// SoA queues; every float array has `cap` entries per component
struct Queues {
    float* ray_o[2]; float* ray_d[2];           // 3*cap each
    float* hit_t; int* hit_prim; float* hit_u; float* hit_v;
    float* thr[2]; uint32_t* id[2]; uint32_t* meta[2]; float* pdf[2];
    float* sh_o; float* sh_d; float* sh_tmax; float* sh_c; uint32_t* sh_id;   // capacity cap*S
    float* L;                                    // 3*cap, indexed by slot id
    uint32_t sh_cap;
};
enum { ST_SAMPLES = 0, ST_EXTEND, ST_SHADE, ST_SHADOW, ST_SHADOW_TRACED, ST_LIT, ST_DRAWS, ST_COUNT };
struct Counters {
    uint32_t n_active[2];
    uint32_t n_shadow;
    uint32_t _pad;
    unsigned long long stats[ST_COUNT];
};

// meta word: draw index [0,16) | bounce [16,24) | is_specular bit 24
APT_D uint32_t pack_meta(uint32_t draw, uint32_t bounce, bool spec) { return (draw & 0xffffu) | (bounce << 16) | (spec ? (1u << 24) : 0u); }

#define BLOCK 256

// LDS carve for the traversal stages (dynamic, sized per scene by the host):
//   [ lds_nodes * 64 B node records | lds_prims * 48 B primitive records | stack_depth * BLOCK ints ]
struct LdsPlan { int lds_nodes, lds_prims, stack_depth; };
extern __shared__ float4 s_dyn[];
APT_D int* carve_lds(const DevBvh& b, const LdsPlan& plan, StagedBvh& out) {
    float4* s_nodes = s_dyn;
    float4* s_prims = s_dyn + 4 * plan.lds_nodes;
    stage_bvh(b, s_nodes, plan.lds_nodes, s_prims, plan.lds_prims, out);
    __syncthreads();
    return reinterpret_cast<int*>(s_prims + 3 * plan.lds_prims) + threadIdx.x;
}

APT_D uint32_t lane_id() { return threadIdx.x & 63u; }
// append `flag` lanes of the wave to a queue counted by *counter; returns this lane's slot
APT_D uint32_t wave_append(bool flag, uint32_t* counter) {
    unsigned long long m = __ballot(flag);
    uint32_t base = 0;
    if (lane_id() == 0 && m) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, 0);
    return base + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
}
APT_D void wave_count(bool flag, unsigned long long* counter) {
    unsigned long long m = __ballot(flag);
    if (lane_id() == 0 && m) atomicAdd(counter, (unsigned long long)__popcll(m));
}
APT_D void wave_sum(uint32_t v, unsigned long long* counter) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane_id() == 0 && v) atomicAdd(counter, (unsigned long long)v);
}

// local pixel -> (global column, row)
APT_D void local_to_global(const Params& p, uint32_t lp, int& i, int& j) {
    int lc = (int)(lp / (uint32_t)p.H);
    j = (int)(lp % (uint32_t)p.H);
    int lb = lc / p.band_width, w = lc % p.band_width;
    i = (lb * p.world + p.rank) * p.band_width + w;
}

// ----------------------------------------------------------------- generate
__global__ void __launch_bounds__(BLOCK) k_generate(Params p, Queues q, Counters* cnt) {
    const uint32_t total = (uint32_t)p.npix * (uint32_t)p.spp_batch;
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t base = blockIdx.x * BLOCK; base < total; base += stride) {
        uint32_t idx = base + threadIdx.x;
        bool valid = idx < total;
        bool alive = false;
        f3 dir = mk3(0.f, 0.f, 1.f);
        uint32_t draws = 0;
        if (valid) {
            uint32_t lp = idx % (uint32_t)p.npix, s = idx / (uint32_t)p.npix;
            int i, j; local_to_global(p, lp, i, j);
            q.L[idx] = 0.f; q.L[p.cap + idx] = 0.f; q.L[2 * p.cap + idx] = 0.f;
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
        uint32_t slot = wave_append(alive, &cnt->n_active[0]);
        if (alive) {
            q.ray_o[0][slot] = p.cam_t[0]; q.ray_o[0][p.cap + slot] = p.cam_t[1]; q.ray_o[0][2 * p.cap + slot] = p.cam_t[2];
            q.ray_d[0][slot] = dir.x; q.ray_d[0][p.cap + slot] = dir.y; q.ray_d[0][2 * p.cap + slot] = dir.z;
            q.thr[0][slot] = 1.f; q.thr[0][p.cap + slot] = 1.f; q.thr[0][2 * p.cap + slot] = 1.f;
            q.id[0][slot] = idx;
            q.meta[0][slot] = pack_meta(draws, 0u, false);
            q.pdf[0][slot] = 1.f;
        }
        wave_count(alive, &cnt->stats[ST_SAMPLES]);
        wave_sum(draws, &cnt->stats[ST_DRAWS]);
    }
}

// ------------------------------------------------------------------- extend
// closest hit for ray queue `cur`; also the stage that recycles the counters of the
// queues nobody reads any more (next-ray queue of this bounce, shadow queue)
__global__ void __launch_bounds__(BLOCK) k_extend(DevScene sc, Params p, Queues q, Counters* cnt, int cur, const uint32_t* n_ptr, LdsPlan plan) {
    StagedBvh bvh;
    int* my_stack = carve_lds(sc.bvh, plan, bvh);
    const uint32_t n = *n_ptr;
    if (blockIdx.x == 0 && threadIdx.x == 0 && cnt) {
        cnt->n_shadow = 0; cnt->n_active[cur ^ 1] = 0;
        cnt->stats[ST_EXTEND] += n;
    }
    const float* ro = q.ray_o[cur]; const float* rd = q.ray_d[cur];
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t idx = blockIdx.x * BLOCK + threadIdx.x; idx < n; idx += stride) {
        f3 o = mk3(ro[idx], ro[p.cap + idx], ro[2 * p.cap + idx]);
        f3 d = mk3(rd[idx], rd[p.cap + idx], rd[2 * p.cap + idx]);
        HitRec rec; rec.t = 1e7f; rec.prim = -1; rec.u = 0.f; rec.v = 0.f;
        traverse<false>(bvh, my_stack, BLOCK, o, d, rec);
        q.hit_t[idx] = rec.t; q.hit_prim[idx] = rec.prim; q.hit_u[idx] = rec.u; q.hit_v[idx] = rec.v;
    }
}

// -------------------------------------------------------------------- shade
APT_D void build_hit(const DevScene& sc, int prim, float t, float u, float v, f3 o, f3 d, Hit& it) {
    it.prim_id = prim; it.min_depth = t;
    it.obj_id = sc.prim_obj[prim];
    if (sc.obj_info[3 * it.obj_id + 2]) {
        // sphere: prims row = (centre, rrr); normal from the hit point (tracer_base.py:217-223)
        f3 c = ld3(sc.precom + 9 * prim);
        it.n_g = normalize((o + d * t) - c);
        it.n_s = it.n_g;
    } else {
        it.n_g = ld3(sc.normals + 3 * prim);
        if (sc.has_vn) {
            const float* vn = sc.vnormals + 9 * prim;
            // interpolated vertex normal, NOT re-normalised (tracer_base.py:228-230)
            it.n_s = (ld3(vn) * (1.f - u - v) + ld3(vn + 3) * u) + ld3(vn + 6) * v;
        } else it.n_s = it.n_g;
    }
}

__global__ void __launch_bounds__(BLOCK) k_shade(DevScene sc, Params p, Queues q, Counters* cnt, int cur, int bounce) {
    const int nxt = cur ^ 1;
    const uint32_t n = cnt->n_active[cur];
    const uint32_t stride = gridDim.x * BLOCK;
    const EmitterGeom geom = {sc.precom, sc.normals, sc.obj_info};
    for (uint32_t base = blockIdx.x * BLOCK; base < n; base += stride) {
        const uint32_t idx = base + threadIdx.x;
        bool alive = idx < n;
        f3 o = splat3(0.f), d = mk3(0.f, 0.f, 1.f), thr = splat3(0.f), hit_point = splat3(0.f);
        uint32_t id = 0;
        float ray_pdf = 1.f;
        bool was_spec = false;
        Philox rng; rng_init(rng, 0u, 0u, 0u, 0u);
        uint32_t draw0 = 0;
        Hit it; it.obj_id = -1; it.prim_id = -1; it.n_s = it.n_g = mk3(1.f, 0.f, 0.f); it.min_depth = 0.f;
        int hit_light = -1;
        float emission_weight = 1.0f;
        DevBxdf bx; bx.type = 1; bx.is_delta = 0; bx.is_bsdf = 0; bx.k_d = bx.k_s = bx.k_g = bx.mean = splat3(0.f); bx.ior = 1.f;
        if (alive) {
            int prim = q.hit_prim[idx];
            if (prim < 0) alive = false;                         // nothing hit: path ends (vanilla_renderer.py:49)
            else {
                o = mk3(q.ray_o[cur][idx], q.ray_o[cur][p.cap + idx], q.ray_o[cur][2 * p.cap + idx]);
                d = mk3(q.ray_d[cur][idx], q.ray_d[cur][p.cap + idx], q.ray_d[cur][2 * p.cap + idx]);
                thr = mk3(q.thr[cur][idx], q.thr[cur][p.cap + idx], q.thr[cur][2 * p.cap + idx]);
                id = q.id[cur][idx];
                uint32_t meta = q.meta[cur][idx];
                ray_pdf = q.pdf[cur][idx];
                was_spec = (meta >> 24) & 1u;
                build_hit(sc, prim, q.hit_t[idx], q.hit_u[idx], q.hit_v[idx], o, d, it);
                bx = sc.bxdf[it.obj_id];
                hit_light = sc.emitter_id[it.obj_id];
                uint32_t lp = id % (uint32_t)p.npix, s = id / (uint32_t)p.npix;
                int gi, gj; local_to_global(p, lp, gi, gj);
                draw0 = meta & 0xffffu;
                rng_init(rng, (uint32_t)(gi * p.H + gj), p.seed, (uint32_t)(p.cnt_base + (int)s + 1), draw0);
                // tail of the previous iteration: emission MIS weight for this hit (vanilla_renderer.py:111-117)
                if (bounce > 0 && p.use_mis) {
                    float e_pdf = 0.0f;
                    if (hit_light >= 0 && bx.is_delta == 0 && !was_spec) e_pdf = emitter_solid_angle_pdf(sc.src[hit_light], it, d);
                    emission_weight = balance(ray_pdf, e_pdf);
                }
                // Russian roulette / cut-off (vanilla_renderer.py:50-57)
                if (p.use_rr) {
                    float mx = max3(thr);
                    if (mx < p.rr_threshold && bounce >= p.rr_bounce_th) {
                        if (rng_float(rng) > mx) alive = false;
                        else thr = thr * (1.f / (mx + 1e-7f));
                    }
                } else if (max3(thr) < 1e-4f) alive = false;
            }
        }
        wave_count(alive, &cnt->stats[ST_SHADE]);
        if (alive) hit_point = d * it.min_depth + o;

        // ---- next-event estimation: one shadow-queue entry per useful light sample
        bool break_flag = false;
        for (int s = 0; s < p.S; s++) {
            bool want = false, cast = false;
            f3 light_dir = splat3(0.f), contrib = splat3(0.f);
            float emitter_d = 0.f;
            if (alive && !break_flag) {
                // sample_light (path_tracer.py:537-554): one int is always drawn
                int ns = sc.n_sources;
                int sidx = pymod(rng_int(rng), ns);
                float emitter_pdf = 1.f / (float)ns;
                bool valid = true;
                if (hit_light >= 0) {
                    if (ns <= 1) valid = false;
                    else {
                        sidx = pymod(rng_int(rng), ns - 1);
                        if (sidx >= hit_light) sidx += 1;
                        emitter_pdf = 1.f / (float)(ns - 1);
                    }
                }
                if (!valid) break_flag = true;
                else {
                    const DevSrc src = sc.src[sidx];
                    f3 shadow_int; float direct_pdf;
                    f3 emit_pos = emitter_sample_hit(src, geom, hit_point, rng, shadow_int, direct_pdf);
                    f3 to_emitter = emit_pos - hit_point;
                    emitter_d = norm(to_emitter);
                    light_dir = to_emitter / emitter_d;
                    cast = true;
                    f3 direct_spec = surface_eval(bx, it, d, light_dir, sc.world_ior, p.two_sides);
                    float mis_w = 1.0f;
                    if (p.use_mis && !(src.bool_bits & 0x01)) {
                        float light_pdf = emitter_pdf * direct_pdf;
                        float bsdf_pdf_v = surface_pdf(bx, it, light_dir, d, sc.world_ior, p.two_sides);
                        mis_w = balance(light_pdf, bsdf_pdf_v);
                    }
                    f3 c = ((direct_spec * shadow_int) * mis_w) / emitter_pdf;
                    contrib = (c * p.inv_S) * thr;
                    want = !(contrib.x == 0.f && contrib.y == 0.f && contrib.z == 0.f);
                }
            }
            wave_count(cast, &cnt->stats[ST_SHADOW]);
            uint32_t slot = wave_append(want, &cnt->n_shadow);
            if (want && slot < q.sh_cap) {
                const uint32_t sc_ = q.sh_cap;
                q.sh_o[slot] = hit_point.x; q.sh_o[sc_ + slot] = hit_point.y; q.sh_o[2 * sc_ + slot] = hit_point.z;
                q.sh_d[slot] = light_dir.x; q.sh_d[sc_ + slot] = light_dir.y; q.sh_d[2 * sc_ + slot] = light_dir.z;
                q.sh_tmax[slot] = emitter_d;
                q.sh_c[slot] = contrib.x; q.sh_c[sc_ + slot] = contrib.y; q.sh_c[2 * sc_ + slot] = contrib.z;
                q.sh_id[slot] = id;
            }
        }

        // ---- emission of the surface we are on, then sample the continuation
        bool cont = false;
        f3 new_d = mk3(0.f, 1.f, 0.f);
        float new_pdf = 1.f;
        bool is_spec = false;
        if (alive) {
            if (hit_light >= 0) {
                f3 emit_int = emitter_eval_le(sc.src[hit_light], hit_point - o, it.n_s);
                if (!(emit_int.x == 0.f && emit_int.y == 0.f && emit_int.z == 0.f)) {
                    f3 add = (emit_int * emission_weight) * thr;
                    q.L[id] += add.x; q.L[p.cap + id] += add.y; q.L[2 * p.cap + id] += add.z;
                }
            }
            f3 spec;
            new_d = surface_sample(bx, it, d, sc.world_ior, p.two_sides, rng, spec, new_pdf, is_spec);
            thr = thr * (spec / new_pdf);
            cont = (bounce + 1) < p.max_bounce;
        }
        wave_sum(rng.draw - draw0, &cnt->stats[ST_DRAWS]);
        uint32_t slot = wave_append(cont, &cnt->n_active[nxt]);
        if (cont) {
            q.ray_o[nxt][slot] = hit_point.x; q.ray_o[nxt][p.cap + slot] = hit_point.y; q.ray_o[nxt][2 * p.cap + slot] = hit_point.z;
            q.ray_d[nxt][slot] = new_d.x; q.ray_d[nxt][p.cap + slot] = new_d.y; q.ray_d[nxt][2 * p.cap + slot] = new_d.z;
            q.thr[nxt][slot] = thr.x; q.thr[nxt][p.cap + slot] = thr.y; q.thr[nxt][2 * p.cap + slot] = thr.z;
            q.id[nxt][slot] = id;
            q.meta[nxt][slot] = pack_meta(rng.draw, (uint32_t)(bounce + 1), is_spec);
            q.pdf[nxt][slot] = new_pdf;
        }
    }
}

// ------------------------------------------------------------------- shadow
__global__ void __launch_bounds__(BLOCK) k_shadow(DevScene sc, Params p, Queues q, Counters* cnt, LdsPlan plan) {
    StagedBvh bvh;
    int* my_stack = carve_lds(sc.bvh, plan, bvh);
    const uint32_t n = min(cnt->n_shadow, q.sh_cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->stats[ST_SHADOW_TRACED] += n;
    const uint32_t stride = gridDim.x * BLOCK, sc_ = q.sh_cap;
    for (uint32_t base = blockIdx.x * BLOCK; base < n; base += stride) {
        uint32_t idx = base + threadIdx.x;
        bool lit = false;
        if (idx < n) {
            f3 o = mk3(q.sh_o[idx], q.sh_o[sc_ + idx], q.sh_o[2 * sc_ + idx]);
            f3 d = mk3(q.sh_d[idx], q.sh_d[sc_ + idx], q.sh_d[2 * sc_ + idx]);
            float dist = q.sh_tmax[idx];
            HitRec rec; rec.t = (dist > 0.0f) ? dist - 1e-4f : 1e7f; rec.prim = -1; rec.u = rec.v = 0.f;
            lit = !traverse<true>(bvh, my_stack, BLOCK, o, d, rec);
            if (lit) {
                uint32_t id = q.sh_id[idx];
                atomicAdd(&q.L[id], q.sh_c[idx]);
                atomicAdd(&q.L[p.cap + id], q.sh_c[sc_ + idx]);
                atomicAdd(&q.L[2 * p.cap + id], q.sh_c[2 * sc_ + idx]);
            }
        }
        wave_count(lit, &cnt->stats[ST_LIT]);
    }
}

// ----------------------------------------------------------------- finalize
// one thread per owned pixel: samples summed in sample order -> bit-reproducible, no atomics
__global__ void __launch_bounds__(BLOCK) k_finalize(Params p, Queues q, float* accum) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t lp = blockIdx.x * BLOCK + threadIdx.x; lp < (uint32_t)p.npix; lp += stride) {
        float r = accum[3 * lp], g = accum[3 * lp + 1], b = accum[3 * lp + 2];
        for (int s = 0; s < p.spp_batch; s++) {
            uint32_t id = (uint32_t)s * (uint32_t)p.npix + lp;
            float cr = q.L[id], cg = q.L[p.cap + id], cb = q.L[2 * p.cap + id];
            r += isnan(cr) ? 0.f : cr; g += isnan(cg) ? 0.f : cg; b += isnan(cb) ? 0.f : cb;
        }
        accum[3 * lp] = r; accum[3 * lp + 1] = g; accum[3 * lp + 2] = b;
    }
}

__global__ void k_divide(const float* accum, float* out, uint32_t n, float cnt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = accum[i] / cnt;        // pixels = color / cnt (vanilla_renderer.py:120)
}

// occlusion test for explicit rays (unit entry point)
__global__ void __launch_bounds__(BLOCK) k_occluded(DevScene sc, uint32_t n, const float* o_, const float* d_, const float* tmax, int* occ, LdsPlan plan) {
    StagedBvh bvh;
    int* my_stack = carve_lds(sc.bvh, plan, bvh);
    for (uint32_t idx = blockIdx.x * BLOCK + threadIdx.x; idx < n; idx += gridDim.x * BLOCK) {
        f3 o = mk3(o_[idx], o_[n + idx], o_[2 * n + idx]), d = mk3(d_[idx], d_[n + idx], d_[2 * n + idx]);
        HitRec rec; rec.t = (tmax[idx] > 0.0f) ? tmax[idx] - 1e-4f : 1e7f; rec.prim = -1; rec.u = rec.v = 0.f;
        occ[idx] = traverse<true>(bvh, my_stack, BLOCK, o, d, rec) ? 1 : 0;
    }
}
__global__ void k_rng_stream(uint32_t pixel, uint32_t seed, uint32_t sample, int n, uint32_t* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        Philox r; rng_init(r, pixel, seed, sample, 0u);
        for (int k = 0; k < n; k++) out[k] = rng_u32(r);
    }
}

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

struct apt_bvh { apt::BvhData data; };

struct apt_scene {
    int device = 0;
    DevScene dev{};
    apt::BvhData bvh;
    DevBuf nodes, prims, normals, vnormals, precom, prim_obj, obj_info, emitter_id, bxdf, src;
    int n_prims = 0, n_objects = 0, n_sources = 0;
};

struct EventPair { hipEvent_t a, b; int kernel; };

struct apt_renderer {
    const apt_scene* scene = nullptr;
    apt_render_cfg cfg{};
    Params par{};
    Queues q{};
    int n_cols = 0, npix = 0, spp_batch = 1;
    int cnt = 0;
    hipStream_t stream = nullptr;
    DevBuf pool, counters, accum, scratch;
    Counters host_counters{};
    int grid_small = 0, grid_trace = 0;
    LdsPlan plan{};
    size_t lds_bytes = 0;
    std::vector<EventPair> pending;
    std::vector<EventPair> free_events;
    double kernel_ms[APT_N_KERNELS] = {0, 0, 0, 0, 0};
    int64_t launches[APT_N_KERNELS] = {0, 0, 0, 0, 0};
    double render_ms = 0.0;
    hipEvent_t ev_r0 = nullptr, ev_r1 = nullptr;
    bool render_pending = false;
};

static int count_device(int* n) {
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess || *n <= 0) { *n = 0; return fail(APT_E_NO_DEVICE, "no HIP device available (this library has no CPU fallback)"); }
    return APT_OK;
}

APT_EXPORT const char* apt_last_error(void) { return g_err.c_str(); }
APT_EXPORT const char* apt_version(void) { return "adapt_mi 0.1 (gfx950 wavefront path tracer)"; }

// ---- BVH build (host only, no device needed)
APT_EXPORT int apt_bvh_build(const float* prims, int32_t n_prims, const int32_t* obj_info, int32_t n_objects, apt_bvh** out) {
    if (!prims || !obj_info || !out || n_prims <= 0 || n_objects <= 0) return fail(APT_E_INVALID, "apt_bvh_build: bad argument");
    apt_bvh* b = new apt_bvh();
    if (apt::build_bvh(prims, n_prims, obj_info, n_objects, b->data) != 0) { delete b; return fail(APT_E_INVALID, "apt_bvh_build: build failed"); }
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
APT_EXPORT void apt_bvh_free(apt_bvh* b) { delete b; }

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
    s->n_prims = N; s->n_objects = O; s->n_sources = S;
    if (apt::build_bvh(d->prims, N, d->obj_info, O, s->bvh) != 0) { delete s; return fail(APT_E_INVALID, "apt_scene_create: BVH build failed"); }

    std::vector<int> prim_obj((size_t)N, 0);
    std::vector<uint8_t> sphere((size_t)N, 0);
    for (int o = 0; o < O; o++)
        for (int k = d->obj_info[3 * o]; k < d->obj_info[3 * o] + d->obj_info[3 * o + 1]; k++) {
            if (k < 0 || k >= N) { delete s; return fail(APT_E_INVALID, "apt_scene_create: obj_info range outside the primitive array"); }
            prim_obj[(size_t)k] = o; sphere[(size_t)k] = d->obj_info[3 * o + 2] != 0;
        }
    // primitive records in BVH order + precom rows (tracer_base.py:117-134)
    std::vector<float> prec((size_t)N * 9), recs((size_t)N * 12, 0.f);
    for (int k = 0; k < N; k++) {
        const float* v = d->prims + 9 * (size_t)k; float* pc = prec.data() + 9 * (size_t)k;
        if (sphere[(size_t)k]) { for (int a = 0; a < 6; a++) pc[a] = v[a]; for (int a = 0; a < 3; a++) pc[6 + a] = v[a]; }
        else for (int a = 0; a < 3; a++) { pc[a] = v[3 + a] - v[a]; pc[3 + a] = v[6 + a] - v[a]; pc[6 + a] = v[a]; }
    }
    for (int slot = 0; slot < N; slot++) {
        int k = s->bvh.prim_order[(size_t)slot];
        const float* v = d->prims + 9 * (size_t)k; const float* pc = prec.data() + 9 * (size_t)k; float* r = recs.data() + 12 * (size_t)slot;
        int32_t kid = k, flag = sphere[(size_t)k] ? 1 : 0;
        if (flag) { r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; }
        else { r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = pc[0]; r[4] = pc[1]; r[5] = pc[2]; r[6] = pc[3]; r[7] = pc[4]; r[8] = pc[5]; }
        memcpy(&r[9], &kid, 4); memcpy(&r[10], &flag, 4);
    }
    std::vector<DevBxdf> bx((size_t)O);
    for (int o = 0; o < O; o++) {
        const int32_t* bi = d->bxdf_i + 4 * o; const float* bf = d->bxdf_f + 13 * o;
        DevBxdf& b = bx[(size_t)o]; memset(&b, 0, sizeof(b));
        b.type = bi[0]; b.is_delta = bi[1]; b.is_bsdf = bi[2];
        b.k_d = mk3(bf[0], bf[1], bf[2]); b.k_s = mk3(bf[3], bf[4], bf[5]); b.k_g = mk3(bf[6], bf[7], bf[8]); b.mean = mk3(bf[9], bf[10], bf[11]); b.ior = bf[12];
    }
    std::vector<DevSrc> sr((size_t)S);
    for (int k = 0; k < S; k++) {
        const int32_t* si = d->src_i + 4 * k; const float* sf = d->src_f + 11 * k;
        DevSrc& e = sr[(size_t)k]; memset(&e, 0, sizeof(e));
        e.type = si[0]; e.bool_bits = si[1]; e.obj_ref_id = si[2];
        e.intensity = mk3(sf[0], sf[1], sf[2]); e.dir = mk3(sf[3], sf[4], sf[5]); e.pos = mk3(sf[6], sf[7], sf[8]); e.inv_area = sf[9]; e.r = sf[10];
        if (e.type == 1 && (e.obj_ref_id < 0 || e.obj_ref_id >= O)) { delete s; return fail(APT_E_INVALID, "apt_scene_create: area emitter is not attached to an object"); }
    }
    std::vector<float> nrm(d->normals, d->normals + (size_t)N * 3);
    std::vector<float> vn((size_t)N * 9, 0.f);
    if (d->v_normals) vn.assign(d->v_normals, d->v_normals + (size_t)N * 9);
    std::vector<int> oi(d->obj_info, d->obj_info + (size_t)O * 3), ei(d->emitter_id, d->emitter_id + (size_t)O);
#define UP(buf, vec) do { hipError_t e_ = upload(s->buf, vec); if (e_ != hipSuccess) { delete s; return fail(APT_E_HIP, std::string("upload " #buf ": ") + hipGetErrorString(e_)); } } while (0)
    UP(nodes, s->bvh.nodes); UP(prims, recs); UP(normals, nrm); UP(vnormals, vn); UP(precom, prec); UP(prim_obj, prim_obj);
    UP(obj_info, oi); UP(emitter_id, ei); UP(bxdf, bx); UP(src, sr);
#undef UP
    DevScene& ds = s->dev;
    ds.bvh.nodes = s->nodes.as<float4>(); ds.bvh.prims = s->prims.as<float4>(); ds.bvh.n_nodes = s->bvh.n_nodes(); ds.bvh.n_prims = N;
    ds.normals = s->normals.as<float>(); ds.vnormals = s->vnormals.as<float>(); ds.precom = s->precom.as<float>();
    ds.prim_obj = s->prim_obj.as<int>(); ds.obj_info = s->obj_info.as<int>(); ds.emitter_id = s->emitter_id.as<int>();
    ds.bxdf = s->bxdf.as<DevBxdf>(); ds.src = s->src.as<DevSrc>();
    ds.n_prims = N; ds.n_objects = O; ds.n_sources = S; ds.has_vn = d->has_vertex_normal; ds.world_ior = d->world_ior;
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

APT_EXPORT int apt_renderer_create(const apt_scene* sc, const apt_render_cfg* cfg, apt_renderer** out) {
    if (!sc || !cfg || !out) return fail(APT_E_INVALID, "apt_renderer_create: null argument");
    apt_render_cfg c = *cfg;
    if (c.width <= 0 || c.height <= 0 || c.max_bounce < 0 || c.num_shadow_ray < 0) return fail(APT_E_INVALID, "apt_renderer_create: bad film / bounce settings");
    if (c.world_size <= 0) { c.world_size = 1; c.rank = 0; }
    if (c.band_width <= 0) c.band_width = c.width;
    if (c.rank < 0 || c.rank >= c.world_size) return fail(APT_E_INVALID, "apt_renderer_create: rank outside world_size");
    if (c.device != sc->device) return fail(APT_E_INVALID, "apt_renderer_create: renderer and scene must live on the same device");
    HIP_TRY(hipSetDevice(c.device));
    apt_renderer* r = new apt_renderer();
    r->scene = sc; r->cfg = c;
    r->n_cols = owned_columns(c);
    // every owned band must be complete except possibly the last: local->global mapping assumes it
    r->npix = r->n_cols * c.height;
    if (r->npix <= 0) { delete r; return fail(APT_E_INVALID, "apt_renderer_create: this rank owns no pixels"); }
    int B = c.spp_per_batch;
    if (B <= 0) { B = (int)((4u << 20) / (uint32_t)r->npix); if (B < 1) B = 1; if (B > 64) B = 64; }
    r->spp_batch = B;
    const size_t cap = (size_t)r->npix * (size_t)B;
    const int S = c.num_shadow_ray;
    const size_t sh_cap = cap * (size_t)(S > 0 ? S : 1);
    Params& p = r->par;
    memcpy(p.cam_r, c.cam_r, sizeof(p.cam_r)); memcpy(p.cam_t, c.cam_t, sizeof(p.cam_t));
    p.inv_focal = c.inv_focal; p.half_w = c.half_w; p.half_h = c.half_h;
    p.W = c.width; p.H = c.height; p.n_cols = r->n_cols; p.npix = r->npix;
    p.band_width = c.band_width; p.rank = c.rank; p.world = c.world_size;
    p.do_crop = c.do_crop; p.sx = c.start_x; p.ex = c.end_x; p.sy = c.start_y; p.ey = c.end_y;
    p.max_bounce = c.max_bounce; p.S = S; p.inv_S = (S > 0) ? 1.f / (float)S : 1.f;
    p.use_rr = c.use_rr; p.use_mis = c.use_mis; p.anti_alias = c.anti_alias; p.stratified = c.stratified; p.two_sides = c.brdf_two_sides;
    p.rr_bounce_th = c.rr_bounce_th; p.rr_threshold = c.rr_threshold; p.seed = c.seed; p.cap = (uint32_t)cap;
    // one pool, carved into the SoA arrays (all 4-byte lanes)
    const size_t words = cap * (6 * 2 + 4 + (3 + 1 + 1 + 1) * 2 + 3) + sh_cap * (3 + 3 + 1 + 3 + 1);
    hipError_t e = r->pool.alloc(words * 4);
    if (e != hipSuccess) { delete r; return fail(APT_E_NOMEM, std::string("queue pool: ") + hipGetErrorString(e)); }
    float* w = r->pool.as<float>();
    auto take = [&](size_t n) { float* x = w; w += n; return x; };
    Queues& q = r->q;
    for (int k = 0; k < 2; k++) { q.ray_o[k] = take(3 * cap); q.ray_d[k] = take(3 * cap); }
    q.hit_t = take(cap); q.hit_prim = (int*)take(cap); q.hit_u = take(cap); q.hit_v = take(cap);
    for (int k = 0; k < 2; k++) { q.thr[k] = take(3 * cap); q.id[k] = (uint32_t*)take(cap); q.meta[k] = (uint32_t*)take(cap); q.pdf[k] = take(cap); }
    q.L = take(3 * cap);
    q.sh_o = take(3 * sh_cap); q.sh_d = take(3 * sh_cap); q.sh_tmax = take(sh_cap); q.sh_c = take(3 * sh_cap); q.sh_id = (uint32_t*)take(sh_cap);
    q.sh_cap = (uint32_t)sh_cap;
    if ((e = r->counters.alloc(sizeof(Counters))) != hipSuccess || (e = r->accum.alloc((size_t)r->npix * 12)) != hipSuccess ||
        (e = r->scratch.alloc((size_t)r->npix * 12)) != hipSuccess) { delete r; return fail(APT_E_NOMEM, std::string("framebuffer: ") + hipGetErrorString(e)); }
    HIP_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    HIP_TRY(hipMemsetAsync(r->counters.p, 0, sizeof(Counters), r->stream));
    HIP_TRY(hipMemsetAsync(r->accum.p, 0, (size_t)r->npix * 12, r->stream));
    HIP_TRY(hipEventCreate(&r->ev_r0)); HIP_TRY(hipEventCreate(&r->ev_r1));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c.device));
    int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // LDS plan: the per-lane stack must cover the tree depth; what is left of a ~40 KiB
    // per-workgroup budget (4 workgroups per CU) stages the top of the tree and, when they
    // all fit, the primitive records.
    {
        const apt::BvhData& bd = sc->bvh;
        LdsPlan pl;
        pl.stack_depth = bd.max_depth + 2;
        size_t stack_b = (size_t)pl.stack_depth * BLOCK * 4;
        size_t budget = 40 * 1024 > stack_b + 8 * 1024 ? 40 * 1024 - stack_b : 8 * 1024;
        int n_nodes = bd.n_nodes(), n_prims = sc->n_prims;
        pl.lds_prims = ((size_t)n_prims * 48 <= budget / 2) ? n_prims : 0;
        size_t left = budget - (size_t)pl.lds_prims * 48;
        pl.lds_nodes = (int)std::min<size_t>((size_t)n_nodes, left / 64);
        r->plan = pl;
        r->lds_bytes = (size_t)pl.lds_nodes * 64 + (size_t)pl.lds_prims * 48 + stack_b;
        if (r->lds_bytes > 160 * 1024) { delete r; return fail(APT_E_INVALID, "apt_renderer_create: BVH too deep for the LDS traversal stack"); }
        int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / r->lds_bytes));
        r->grid_trace = cus * per_cu;
        if (r->lds_bytes > 64 * 1024) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_extend, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
            HIP_TRY(hipFuncSetAttribute((const void*)k_shadow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
            HIP_TRY(hipFuncSetAttribute((const void*)k_occluded, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r->lds_bytes));
        }
    }
    r->grid_small = cus * 8;       // streaming stages: 8 x 256-thread workgroups per CU
    HIP_TRY(hipStreamSynchronize(r->stream));
    *out = r;
    return APT_OK;
}
APT_EXPORT void apt_renderer_destroy(apt_renderer* r) {
    if (!r) return;
    (void)hipSetDevice(r->cfg.device);
    if (r->stream) { (void)hipStreamSynchronize(r->stream); (void)hipStreamDestroy(r->stream); }
    for (auto& ev : r->pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto& ev : r->free_events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    if (r->ev_r0) (void)hipEventDestroy(r->ev_r0);
    if (r->ev_r1) (void)hipEventDestroy(r->ev_r1);
    delete r;
}

static int grid_for(size_t n, int cap_blocks) {
    size_t b = (n + BLOCK - 1) / BLOCK;
    if (b < 1) b = 1;
    return (int)(b < (size_t)cap_blocks ? b : (size_t)cap_blocks);
}

struct LaunchTimer {      // brackets one kernel launch with events when profiling is on
    apt_renderer* r; int kernel; EventPair ev{}; bool on;
    LaunchTimer(apt_renderer* r_, int k) : r(r_), kernel(k), on(r_->cfg.profile != 0) {
        r->launches[k]++;
        if (!on) return;
        if (!r->free_events.empty()) { ev = r->free_events.back(); r->free_events.pop_back(); }
        else { (void)hipEventCreate(&ev.a); (void)hipEventCreate(&ev.b); }
        ev.kernel = k;
        (void)hipEventRecord(ev.a, r->stream);
    }
    ~LaunchTimer() { if (on) { (void)hipEventRecord(ev.b, r->stream); r->pending.push_back(ev); } }
};

static int resolve_events(apt_renderer* r) {
    HIP_TRY(hipStreamSynchronize(r->stream));
    for (auto& ev : r->pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        r->kernel_ms[ev.kernel] += ms;
        r->free_events.push_back(ev);
    }
    r->pending.clear();
    if (r->render_pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r->ev_r0, r->ev_r1));
        r->render_ms += ms; r->render_pending = false;
    }
    return APT_OK;
}

APT_EXPORT int apt_render(apt_renderer* r, int32_t n_spp) {
    if (!r || n_spp < 0) return fail(APT_E_INVALID, "apt_render: bad argument");
    HIP_TRY(hipSetDevice(r->cfg.device));
    if (r->render_pending) { if (int rc = resolve_events(r)) return rc; }
    const DevScene& sc = r->scene->dev;
    Counters* cnt = r->counters.as<Counters>();
    HIP_TRY(hipEventRecord(r->ev_r0, r->stream));
    int done = 0;
    while (done < n_spp) {
        const int B = (n_spp - done < r->spp_batch) ? (n_spp - done) : r->spp_batch;
        Params p = r->par; p.cnt_base = r->cnt; p.spp_batch = B;
        const size_t total = (size_t)r->npix * (size_t)B;
        HIP_TRY(hipMemsetAsync(cnt, 0, 16, r->stream));                 // queue counters only, stats keep accumulating
        { LaunchTimer t(r, 0); hipLaunchKernelGGL(k_generate, dim3(grid_for(total, r->grid_small)), dim3(BLOCK), 0, r->stream, p, r->q, cnt); }
        int cur = 0;
        for (int b = 0; b < p.max_bounce; b++) {
            { LaunchTimer t(r, 1); hipLaunchKernelGGL(k_extend, dim3(grid_for(total, r->grid_trace)), dim3(BLOCK), r->lds_bytes, r->stream, sc, p, r->q, cnt, cur, (const uint32_t*)&cnt->n_active[cur], r->plan); }
            { LaunchTimer t(r, 2); hipLaunchKernelGGL(k_shade, dim3(grid_for(total, r->grid_small)), dim3(BLOCK), 0, r->stream, sc, p, r->q, cnt, cur, b); }
            if (p.S > 0) { LaunchTimer t(r, 3); hipLaunchKernelGGL(k_shadow, dim3(grid_for(total * (size_t)p.S, r->grid_trace)), dim3(BLOCK), r->lds_bytes, r->stream, sc, p, r->q, cnt, r->plan); }
            cur ^= 1;
        }
        { LaunchTimer t(r, 4); hipLaunchKernelGGL(k_finalize, dim3(grid_for((size_t)r->npix, r->grid_small)), dim3(BLOCK), 0, r->stream, p, r->q, r->accum.as<float>()); }
        HIP_TRY(hipGetLastError());
        r->cnt += B; done += B;
    }
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
    Counters h;
    HIP_TRY(hipMemcpy(&h, r->counters.p, sizeof(Counters), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    out->n_samples = (int64_t)h.stats[ST_SAMPLES]; out->n_extend = (int64_t)h.stats[ST_EXTEND]; out->n_shade = (int64_t)h.stats[ST_SHADE];
    out->n_shadow = (int64_t)h.stats[ST_SHADOW]; out->n_shadow_traced = (int64_t)h.stats[ST_SHADOW_TRACED]; out->n_lit = (int64_t)h.stats[ST_LIT];
    out->n_draws = (int64_t)h.stats[ST_DRAWS];
    for (int k = 0; k < APT_N_KERNELS; k++) { out->launches[k] = r->launches[k]; out->kernel_ms[k] = r->kernel_ms[k]; }
    out->render_ms = r->render_ms;
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
    hipLaunchKernelGGL(k_extend, dim3(grid_for((size_t)n, r->grid_trace)), dim3(BLOCK), r->lds_bytes, r->stream, r->scene->dev, r->par, r->q, (Counters*)nullptr, 0,
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
    hipLaunchKernelGGL(k_occluded, dim3(grid_for((size_t)n, r->grid_trace)), dim3(BLOCK), r->lds_bytes, r->stream, r->scene->dev, (uint32_t)n,
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
