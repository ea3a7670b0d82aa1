// Volumetric path tracer stages (homogeneous participating media and grid volumes).
//
// Replaces VolumeRenderer.render of the reference (renderer/vpt.py:145-258) on the same wavefront skeleton as the surface path
// tracer: k_generate and k_extend are shared; k_vevent + k_vshade_ev are the loop body between two closest-hit queries (Russian roulette,
// free path sampling, null-surface pass-through | light sampling, emission, phase-function / surface scattering) and k_vshadow is
// track_ray (vpt.py:99-138): the transmittance walk of a light sample through null surfaces and media, up to seven closest-hit
// queries per sample.  Media: the world's and those attached to BSDF objects (bxdf/medium.py:71-125), phase functions H-G,
// multi-H-G and Rayleigh (bxdf/phase.py, sampler/phase_sampling.py), and one RGB grid volume (bxdf/volume.py: delta tracking for
// free paths, ratio tracking for light samples).
#pragma once
#include "shade_stage.hpp"
#ifndef APT_VSHADOW_WAVES
#define APT_VSHADOW_WAVES 5
#endif
// workgroup size of the tiled transmittance walk: its closest-hit lists cost 8 B per thread and object in LDS, so on scenes of 8-9
// objects a 512-thread tile leaves room for two workgroups per CU only; 256-thread tiles: V1 walk 38.2 -> 35.3 ms, V2 52.7 -> 50.4 ms
#ifndef APT_VSHADOW_NT
#define APT_VSHADOW_NT 256
#endif
#define VSHADOW_NT(MODE) ((MODE) == 2 ? APT_VSHADOW_NT : BLOCK)

APT_D float med_random_rgb(Philox& r, f3 v) {                      // general_sampling.py:17-27
    const int idx = pymod(rng_int(r), 3);
    const float res = (idx == 0) ? v.x : ((idx == 1) ? v.y : v.z);
    return fmaxf(res, 1e-5f);
}
APT_D f3 exp_neg(f3 u_e, float d) { return mk3(expf(-u_e.x * d), expf(-u_e.y * d), expf(-u_e.z * d)); }
APT_D float sum3(f3 a) { return (a.x + a.y) + a.z; }
// Medium.sample_mfp, medium.py:89-108
APT_D bool medium_sample_mfp(const DevMedium& m, float max_depth, Philox& r, float& t, f3& beta) {
    const float random_ue = med_random_rgb(r, m.u_e);
    float sample_t = -logf(1.f - rng_float(r)) / random_ue;
    bool is_mi = false;
    if (sample_t >= max_depth) {
        sample_t = max_depth;
        const f3 tr = exp_neg(m.u_e, max_depth);
        float pdf = sum3(tr) / 3.f;
        pdf = (pdf > 0.f) ? pdf : 1.f;
        beta = tr / pdf;
    } else {
        is_mi = true;
        const f3 tr = exp_neg(m.u_e, sample_t);
        float pdf = sum3(m.u_e * tr) / 3.f;
        pdf = (pdf > 0.f) ? pdf : 1.f;
        beta = (tr * m.u_s) / pdf;
    }
    t = sample_t;
    return is_mi;
}
// bxdf/phase.py:21-31
APT_D float phase_hg(float cos_theta, float g) {
    const float g2 = g * g;
    const float denom = (1.f + g2) - (2.f * g) * cos_theta;
    return (((1.f - g2) / (sqrtf(denom) * denom)) * 0.5f) * APT_INV_2PI;
}
APT_D float phase_rayleigh(float cos_theta) { return (float)(0.375 * ((1.0 / 3.14159265358979323846) * 0.5)) * (1.f + cos_theta * cos_theta); }
// sampler/phase_sampling.py:16-42
APT_D f3 sample_hg(Philox& r, float g, float& cos_theta) {
    if (fabsf(g) < 1e-4f) cos_theta = 1.f - 2.f * rng_float(r);
    else {
        const float g2 = g * g;
        const float sqr_term = (1.f - g2) / ((1.f + g) - (2.f * g) * rng_float(r));
        cos_theta = ((1.f + g2) - sqr_term * sqr_term) / (2.f * g);
    }
    const float sin_theta = sqrtf(fmaxf(0.f, 1.f - cos_theta * cos_theta));
    const float phi = APT_2PI * rng_float(r);
    return polar_dir(cos_theta, sin_theta, phi);
}
APT_D f3 sample_rayleigh(Philox& r, float& cos_theta) {
    const float rd = 2.f * rng_float(r) - 1.f;
    const float u = -apt_pow(2.f * rd + sqrtf((4.f * rd) * rd + 1.f), (float)(1.0 / 3.0));
    cos_theta = fminf(fmaxf(u - 1.f / u, -1.f), 1.f);
    const float sin_theta = sqrtf(fmaxf(0.f, 1.f - cos_theta * cos_theta));
    const float phi = APT_2PI * rng_float(r);
    return polar_dir(cos_theta, sin_theta, phi);
}
// PhaseFunction.sample_p / eval_p, phase.py:39-84
APT_D f3 phase_sample_p(const DevMedium& m, f3 incid, Philox& r, float& p) {
    f3 dir = incid; p = 1.f; float cos_t = 0.f;
    if (m.type == 0) { dir = sample_hg(r, m.par.x, cos_t); p = phase_hg(cos_t, m.par.x); }
    else if (m.type == 1) {
        const float eps = rng_float(r);
        const float g = (eps < m.pdf.x) ? m.par.x : ((eps < m.pdf.x + m.pdf.y) ? m.par.y : m.par.z);
        dir = sample_hg(r, g, cos_t); p = phase_hg(cos_t, g);
    } else if (m.type == 2) { dir = sample_rayleigh(r, cos_t); p = phase_rayleigh(cos_t); }
    return dir;
}
APT_D float phase_eval_p(const DevMedium& m, f3 ray_in, f3 ray_out) {
    float p = 1.f; const float cos_theta = -dot(ray_in, ray_out);
    if (m.type == 0) p = phase_hg(cos_theta, m.par.x);
    else if (m.type == 1) {
        p = phase_hg(cos_theta, m.par.x) * m.pdf.x + phase_hg(cos_theta, m.par.y) * m.pdf.y;
        if (m.pdf.y > 1e-4f) p += phase_hg(cos_theta, m.par.z) * m.pdf.z;
    } else if (m.type == 2) p = phase_rayleigh(cos_theta);
    return p;
}
APT_D bool vpt_is_scattering(const DevScene& sc, int idx) { return idx >= 0 && sc.bxdf[idx].is_bsdf && sc.med[idx].type >= 0; }    // path_tracer.py:528-535
APT_D bool vpt_non_null(const DevScene& sc, int idx) { return !(idx >= 0 && sc.bxdf[idx].is_bsdf) || sc.bxdf[idx].type >= 0; }     // vpt.py:64-70
// VolumeRenderer.world_bound_time, vpt.py:140-143 (max of the slab pairs ignores NaN, the min over the axes propagates it)
APT_D float world_bound_time(const Params& p, f3 o, f3 d) {
    const f3 lo = mk3(p.w_min[0], p.w_min[1], p.w_min[2]), hi = mk3(p.w_max[0], p.w_max[1], p.w_max[2]);
    const f3 t_min = (lo - o) / d, t_max = (hi - o) / d;
    const f3 m = mk3(fmaxf(t_min.x, t_max.x), fmaxf(t_min.y, t_max.y), fmaxf(t_min.z, t_max.z));
    float r = m.x;
    if (!isnan(r) && (isnan(m.y) || m.y < r)) r = m.y;
    if (!isnan(r) && (isnan(m.z) || m.z < r)) r = m.z;
    return r;
}

// ------------------------------------------------------------- grid volume, bxdf/volume.py:248-463 (RGB volumes)
APT_D float max_np(f3 a) { return (isnan(a.x) || isnan(a.y) || isnan(a.z)) ? NAN : max3(a); }       // Vector.max() / .min() propagate NaN
APT_D float min_np(f3 a) { return (isnan(a.x) || isnan(a.y) || isnan(a.z)) ? NAN : min3(a); }
APT_D bool vol_intersect(const DevVolume& vo, f3 o, f3 d, float max_t, float& near_t, float& far_t) {        // volume.py:271-285
    const f3 inv_dir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const f3 t1s = (vo.mini - o) * inv_dir, t2s = (vo.maxi - o) * inv_dir;
    near_t = fmaxf(0.f, max_np(min3v(t1s, t2s))) + 1e-5f;
    far_t = fminf(max_t, min_np(max3v(t1s, t2s))) - 1e-5f;
    return near_t < far_t && far_t > 0.f;
}
APT_D f3 vol_to_local(const DevVolume& vo, f3 p) { return mk3(dot(vo.inv_r0, p), dot(vo.inv_r1, p), dot(vo.inv_r2, p)); }
APT_D float pick3(f3 a, int ch) { return (ch == 0) ? a.x : ((ch == 1) ? a.y : a.z); }
// stochastic nearest-voxel lookup of one channel (volume.py:307-314 + rgb_select)
APT_D float vol_density(const DevVolume& vo, f3 index, f3 u, int ch) {
    const int ix = (int)floorf(index.x + (u.x - 0.5f)), iy = (int)floorf(index.y + (u.y - 0.5f)), iz = (int)floorf(index.z + (u.z - 0.5f));
    if (ix >= 0 && iy >= 0 && iz >= 0 && ix < vo.xres && iy < vo.yres && iz < vo.zres)
        return vo.grid[3 * (((size_t)iz * vo.yres + iy) * vo.xres + ix) + ch];
    return 0.f;
}
// wavelength channel by throughput x majorant pdf (volume.py:352-377, 410-432)
APT_D int vol_pick_channel(const DevVolume& vo, f3 thp, Philox& r, float& pdf) {
    f3 pdfs = thp * vo.pdf;
    pdfs = pdfs / sum3(pdfs);
    const float val = rng_float(r);
    if (val <= pdfs.x) { pdf = pdfs.x; return 0; }
    if (val <= pdfs.x + pdfs.y) { pdf = pdfs.y; return 1; }
    pdf = pdfs.z; return 2;
}
APT_D f3 channel_vec(int ch, float v) { return mk3(ch == 0 ? v : 0.f, ch == 1 ? v : 0.f, ch == 2 ? v : 0.f); }
// GridVolume.sample_mfp: delta tracking (volume.py:295-305, 346-397); returns the collision distance or -1
APT_D float vol_sample_mfp(const DevVolume& vo, f3 ray_o, f3 ray_d, f3 thp, float max_t, Philox& r, f3& beta) {
    beta = splat3(1.f);
    float near_t, far_t;
    if (!vol_intersect(vo, ray_o, ray_d, max_t, near_t, far_t)) return -1.f;
    const f3 ol = vol_to_local(vo, ray_o - vo.trans), dl = vol_to_local(vo, ray_d);
    float pdf; const int ch = vol_pick_channel(vo, thp, r, pdf);
    const float albedo = pick3(vo.albedo, ch), inv_maj = 1.0f / pick3(vo.majorant, ch);
    float Tr = 1.0f, hit_t = -1.f;
    float t = near_t - logf(1.0f - rng_float(r)) * inv_maj;
    while (t < far_t) {
        const float u0 = rng_float(r), u1 = rng_float(r), u2 = rng_float(r);
        const float n_t = vol_density(vo, ol + dl * t, mk3(u0, u1, u2), ch);
        if (rng_float(r) < n_t * inv_maj) { Tr *= albedo; hit_t = t; break; }
        t -= logf(1.0f - rng_float(r)) * inv_maj;
    }
    beta = (vo.type == 2) ? channel_vec(ch, Tr / pdf) : splat3(Tr);
    return hit_t;
}
// GridVolume.transmittance: ratio tracking with roulette (volume.py:283-293, 399-463)
APT_D f3 vol_transmittance(const DevVolume& vo, f3 ray_o, f3 ray_d, f3 thp, float max_t, Philox& r) {
    float near_t, far_t;
    if (!vol_intersect(vo, ray_o, ray_d, max_t, near_t, far_t)) return splat3(1.f);
    const f3 ol = vol_to_local(vo, ray_o - vo.trans), dl = vol_to_local(vo, ray_d);
    float pdf; const int ch = vol_pick_channel(vo, thp, r, pdf);
    const float inv_maj = 1.0f / pick3(vo.majorant, ch);
    float Tr = 1.0f, t = near_t;
    for (;;) {
        t -= logf(1.0f - rng_float(r)) * inv_maj;
        if (t >= far_t) break;
        const float u0 = rng_float(r), u1 = rng_float(r), u2 = rng_float(r);
        Tr *= fmaxf(0.0f, 1.0f - vol_density(vo, ol + dl * t, mk3(u0, u1, u2), ch) * inv_maj);
        if (Tr < 0.1f) {
            if (rng_float(r) >= Tr) { Tr = 0.0f; break; }
            Tr = 1.0f;
        }
    }
    return (vo.type == 2) ? channel_vec(ch, Tr / pdf) : splat3(Tr);
}

// ------------------------------------------------------------------- events
// One iteration of the while-loop of vpt.py:161-253 in two stages, sorted by EVENT in between.  (Rounds 1-4 ran the whole loop body as ONE
// kernel, k_vshade: roulette, free-path sampling, null-surface pass-through, light sampling with the phase function AND every surface model
// of its class, two ways to scatter - 170-240 VGPRs, two waves per SIMD, and a wave's lanes in different branches: a medium event next to a
// surface hit next to a ray that only crosses the fog cube's skin.  Measured against it, round 5: V1 1 051 -> 1 241, V2 665 -> 995, V3 638 ->
// 990 Msamples/s; the one-kernel form is gone.)  The bounce counter lives in the path's meta word (a null-surface pass-through re-queues the
// path without counting a bounce) and the float slot that carries ray_pdf in the surface tracer carries emission_weight here (vpt.py:247-253
// computes it at the END of an iteration, from the interaction being left).  What a path does in an iteration is decided by steps 1-3
// alone, before anything is shaded:
//   k_vevent      steps 1-3 for every ray of the queue, given its closest hit (unsorted extend): Russian roulette, the hit or the world
//                 box, free-path sampling (+ delta tracking through a grid volume).  A path that ends is dropped; a path that crosses a
//                 null surface is re-queued at once (no bounce counted); every other path is an EVENT - a medium interaction, or a hit on a
//                 surface of class c - and its 64-byte record joins that event's queue (Queues::cq planes, as the surface tracer's class
//                 queues: A = origin + event distance, B = direction + primitive, C = throughput x beta + path id, D = meta, emission weight,
//                 u, v).  No surface model, no emitter, no phase function in it.
//   k_vshade_ev   steps 4-6 for one event queue: <MI = 1> the medium kernel (phase function, no surface model), <MI = 0, class mask> one
//                 surface class (no phase function, no free-path code): light sampling, emission, scattering, the next hit's emission weight.
// The draws are taken in the reference's order (the draw index travels in the record).  BM / SM: material and emitter masks as in k_shade (code
// for absent models is compiled out; the all-models kernel also carries the image-texture lookup); VOL: the scene holds a grid volume (delta
// tracking in the free-path step, ratio tracking inside the light sampling).
// meta word of an event record: draw index [0,23) | bounce [23,31) | bit 31: the interaction is a grid-volume collision
// A medium event in front of a SPHERE keeps the sphere's own hit distance in the record's u slot (build_hit needs it for the normal
// upstream computed before the free path overwrote min_depth, and a sphere has no barycentrics).
template <int VOL>
__global__ void __launch_bounds__(BLOCK) k_vevent(DevScene sc, Params p, Queues q, Counters* cnt, int cur, int single_class) {
    const int nxt = cur ^ 1;
    const SubLoop sl = sub_loop(p.nq);
    const uint32_t n = cnt->n_active[cur][sl.q * CNT_PAD];
    const uint32_t qbase = (uint32_t)sl.q * p.subcap;
    uint32_t* next_counter = &cnt->n_active[nxt][sl.q * CNT_PAD];
    const DevMedium* world = sc.med + sc.n_objects;           // wave-uniform address: scalar loads
    const bool world_scat = world->type >= 0;
    const int med_class = q.n_classes - 1;                    // the last event queue takes the medium interactions
    __shared__ uint32_t s_draws[BLOCK / 64];
    if (lane_id() == 0) s_draws[threadIdx.x >> 6] = 0;
    for (uint32_t base = sl.first; base < n; base += sl.stride) {
        const uint32_t pos = base + threadIdx.x;
        const uint32_t io = (qbase + pos) << 2;
        bool alive = pos < n, is_mi = false, vol_event = false, in_free = true, pass = false;
        f3 o = splat3(0.f), d = mk3(0.f, 0.f, 1.f), thr = splat3(0.f);
        uint32_t id = 0, draw0 = 0, bounce = 0;
        float emission_weight = 1.f, t_surface = 0.f, hu = 0.f, hv = 0.f;
        int prim = -1, ev = -1;
        Philox rng; rng_init(rng, 0u, 0u, 0u, 0u);
        Hit it; it.obj_id = -1; it.prim_id = -1; it.n_s = it.n_g = mk3(1.f, 0.f, 0.f); it.min_depth = 0.f;
        if (alive) {
            o = ld3q(q.ray_o[cur], p.cap, io);
            d = ld3q(q.ray_d[cur], p.cap, io);
            thr = ld3q(q.thr[cur], p.cap, io);
            id = ldq(q.id[cur], io);
            const uint32_t meta = ldq(q.meta[cur], io);
            emission_weight = ldq(q.pdf[cur], io);
            bounce = (meta >> 23) & 0xffu; draw0 = meta & 0x7fffffu;
            const uint32_t lp = id & ((1u << p.pix_bits) - 1u), s = id >> p.pix_bits;
            rng_init(rng, (p.world == 1) ? lp : ldq(p.pix_key, lp << 2), p.seed, (uint32_t)(p.cnt_base + (int)s + 1), draw0);
            // Step 1: Russian roulette / cut-off BEFORE the intersection is looked at (vpt.py:164-172)
            if (p.use_rr) {
                const float mx = max3(thr);
                if (mx < p.rr_threshold && (int)bounce >= p.rr_bounce_th) {
                    if (rng_float(rng) > mx) alive = false;
                    else thr = thr * (1.f / (mx + 1e-7f));
                }
            } else if (max3(thr) < 1e-5f) alive = false;
            if (alive) {
                // Step 2: the hit, or the far side of the world box when the world itself scatters (vpt.py:173-181)
                prim = ldq(q.hit_prim, io);
                if (prim < 0) {
                    if (!world_scat && !VOL) alive = false;
                    else { it.min_depth = world_bound_time(p, o, d); in_free = true; }
                } else {
                    hu = ldq(q.hit_u, io); hv = ldq(q.hit_v, io);
                    { int l_; f3 kd_; build_hit_rec(sc, sc.prim_shade[2 * prim], sc.prim_shade[2 * prim + 1], prim, ldq(q.hit_t, io), hu, hv, o, d, it, l_, kd_, false); }      // (object, geometric normal, distance: the shading normal is the event kernel's business)
                    in_free = dot(it.n_g, d) < 0.f;
                    t_surface = it.min_depth;
                }
            }
            if (alive) {
                // Step 3: free-path sampling in the medium the segment crosses (vpt.py:72-97,184)
                f3 beta = splat3(1.f);
                const bool world_valid = in_free && world_scat;
                const float depth0 = it.min_depth;
                if (world_valid || vpt_is_scattering(sc, it.obj_id)) {
                    float mfp = it.min_depth;
                    if (world_valid) is_mi = medium_sample_mfp(*world, it.min_depth, rng, mfp, beta);
                    else if (!in_free) is_mi = medium_sample_mfp(sc.med[it.obj_id], it.min_depth, rng, mfp, beta);
                    it.min_depth = mfp;
                }
                if (VOL) {                                                  // a grid-volume event overrides the homogeneous one (vpt.py:91-96)
                    f3 vb; const float vt = vol_sample_mfp(sc.vol, o, d, thr, depth0, rng, vb);
                    if (vt > 0.f) { is_mi = true; vol_event = true; it.min_depth = vt; beta = vb; }
                }
                if (it.obj_id < 0 && !is_mi) alive = false;                 // left the world box
                else {
                    thr = thr * beta;
                    if (!is_mi && !vpt_non_null(sc, it.obj_id)) pass = true;    // null surface: walk on, no bounce counted (vpt.py:189-191)
                    else ev = is_mi ? med_class : (single_class ? 0 : sc.prim_class[prim]);
                }
            }
        }
        if (rng.draw != draw0) atomicAdd(&s_draws[threadIdx.x >> 6], rng.draw - draw0);
        if (pass && rng.draw > 0x7fffffu) atomicAdd(&cnt->stats[sl.q][ST_OVERFLOW], 1ull);      // counted where a next-ray record is written with the truncated index: here for pass-throughs, in the event kernels for continuations (once per record, not once per kernel the path crosses)
        // the pass-throughs go straight back into the ray queue, the events into their queues (all queue tails with one atomic instruction)
        const uint32_t npos = wave_append(pass, next_counter);
        if (pass) {
            const uint32_t so = (qbase + npos) << 2;
            st3q(q.ray_o[nxt], p.cap, so, d * it.min_depth + o);
            st3q(q.ray_d[nxt], p.cap, so, d);
            st3q(q.thr[nxt], p.cap, so, thr);
            stq(q.id[nxt], so, id);
            stq(q.meta[nxt], so, (rng.draw & 0x7fffffu) | (bounce << 23));
            stq(q.pdf[nxt], so, emission_weight);
        }
        const TrAppend ap = tr_append_issue(ev, q.n_classes, &cnt->n_cls[0][sl.q * CNT_PAD], APT_MAX_NQ * CNT_PAD);
        const uint32_t epos = tr_append_pos(ap, ev);
        if (ev >= 0) {
            const bool sphere_mi = is_mi && prim >= 0 && __float_as_int(sc.prim_shade[2 * prim].w) < 0;
            cq_store(q, (uint32_t)ev * p.cap + qbase + epos, o, d, thr, id, (rng.draw & 0x7fffffu) | (bounce << 23) | (vol_event ? 0x80000000u : 0u), emission_weight,
                     it.min_depth, prim, sphere_mi ? t_surface : hu, hv);
        }
    }
    if (lane_id() == 0 && s_draws[threadIdx.x >> 6]) atomicAdd(&cnt->stats[sl.q][ST_DRAWS], (unsigned long long)s_draws[threadIdx.x >> 6]);
}

template <int BM, int SM, int VOL, int MI>
APT_D void vshade_ev_body(const DevScene& sc, const Params& p, const Queues& q, Counters* cnt, int cls, int cur) {
    constexpr bool TEX = (BM == APT_BX_ALL);
    const int nxt = cur ^ 1;
    const SubLoop sl = sub_loop(p.nq);
    const uint32_t n = cnt->n_cls[cls][sl.q * CNT_PAD];
    const uint32_t qbase = (uint32_t)sl.q * p.subcap, sh_qbase = (uint32_t)sl.q * q.sh_subcap, in_base = (uint32_t)cls * p.cap + qbase;
    uint32_t* next_counter = &cnt->n_active[nxt][sl.q * CNT_PAD];
    uint32_t* shadow_counter = &cnt->n_shadow[sl.q * CNT_PAD];
    const EmitterGeom geom = {sc.precom, sc.normals, sc.obj_info};
    const DevMedium* world = sc.med + sc.n_objects;
    uint32_t t_shade = 0, t_shadow = 0, t_poison = 0;
    __shared__ uint32_t s_draws[BLOCK / 64];
    if (lane_id() == 0) s_draws[threadIdx.x >> 6] = 0;
    for (uint32_t base = sl.first; base < n; base += sl.stride) {
        const uint32_t pos = base + threadIdx.x;
        const bool shade = pos < n;
        const uint32_t so16 = (in_base + (shade ? pos : n - 1u)) << 4;
        const float4 ra = ldq(q.cq[0], so16), rb = ldq(q.cq[1], so16), rc = ldq(q.cq[2], so16), rd = ldq(q.cq[3], so16);
        const f3 o = mk3(ra.x, ra.y, ra.z), d = mk3(rb.x, rb.y, rb.z);
        f3 thr = mk3(rc.x, rc.y, rc.z);
        const int prim = __float_as_int(rb.w);
        const uint32_t id = __float_as_uint(rc.w), meta = __float_as_uint(rd.x);
        float emission_weight = rd.y;
        uint32_t bounce = (meta >> 23) & 0xffu; const uint32_t draw0 = meta & 0x7fffffu; const bool vol_event = (meta >> 31) != 0u;
        const uint32_t lp = id & ((1u << p.pix_bits) - 1u), s_ = id >> p.pix_bits;
        const uint32_t l_off = (s_ * (uint32_t)p.npix + lp) << 2;
        Philox rng; rng_init(rng, (p.world == 1) ? lp : ldq(p.pix_key, lp << 2), p.seed, (uint32_t)(p.cnt_base + (int)s_ + 1), draw0);
        Hit it; it.obj_id = -1; it.prim_id = -1; it.n_s = it.n_g = mk3(1.f, 0.f, 0.f); it.min_depth = 0.f;
        int hit_light = -1, rec_light = -1;
        DevBxdf bx; bx.type = 1; bx.is_delta = 0; bx.is_bsdf = 0; bx.k_d = bx.k_s = bx.k_g = bx.mean = splat3(0.f); bx.ior = 1.f;
        bool in_free = true;
        if (prim >= 0) {
            f3 rec_kd;
            // (a medium event in front of a sphere: the sphere's own hit distance rides in the u slot, see k_vevent)
            const bool sphere = __float_as_int(sc.prim_shade[2 * prim].w) < 0;
            build_hit(sc, prim, (MI && sphere) ? rd.z : ra.w, rd.z, rd.w, o, d, it, rec_light, rec_kd);
            in_free = dot(it.n_g, d) < 0.f;
            bx = sc.bxdf[it.obj_id];
        }
        it.min_depth = ra.w;                                               // the event's distance: the hit, or the sampled free path in front of it
        const DevMedium* med = (in_free || it.obj_id < 0) ? world : sc.med + it.obj_id;     // what a medium event is evaluated with (path_tracer.py:466-470)
        const f3 hit_point = d * it.min_depth + o;
        if (!MI) {
            hit_light = rec_light;
            f3 tx;
            if (TEX && sc.tex_i != nullptr && get_uv_item(sc, 0, it.obj_id, it.prim_id, rd.z, rd.w, tx)) bx.k_d = tx;   // vpt.py:199
        }
        t_shade += wave_count(shade);

        // ---- Step 4: light sampling; the transmittance along the sample is k_vshadow's job
        bool break_flag = false;
        DevSrc src_only;
        if (sc.n_sources == 1) src_only = ld_src_uniform(sc.src);
        for (int s = 0; s < p.S; s++) {
            bool want = false, sampled = false, poisoned = false;
            f3 light_dir = splat3(0.f), contrib = splat3(0.f);
            float emitter_d = 0.f;
            if (shade && !break_flag) {
                const int ns = sc.n_sources;
                int sidx = rng_int(rng);
                sidx = (ns == 1) ? 0 : pymod(sidx, ns);
                float emitter_pdf = p.inv_ns;
                bool valid = true;
                if (hit_light >= 0) {
                    if (ns <= 1) valid = false;
                    else {
                        sidx = rng_int(rng);
                        sidx = (ns == 2) ? 0 : pymod(sidx, ns - 1);
                        if (sidx >= hit_light) sidx += 1;
                        emitter_pdf = p.inv_ns1;
                    }
                }
                if (!valid) break_flag = true;
                else {
                    const DevSrc src = (ns == 1) ? src_only : sc.src[sidx];
                    f3 shadow_int; float direct_pdf;
                    const f3 emit_pos = emitter_sample_hit<SM>(src, geom, hit_point, rng, shadow_int, direct_pdf);
                    const f3 to_emitter = emit_pos - hit_point;
                    emitter_d = norm(to_emitter);
                    light_dir = to_emitter / emitter_d;
                    sampled = true;
                    if (VOL) shadow_int = shadow_int * vol_transmittance(sc.vol, hit_point, light_dir, thr, emitter_d, rng);    // track_ray's first step (vpt.py:107-108): draws from the path's stream, here and now
                    f3 direct_spec;
                    if (MI) direct_spec = splat3(phase_eval_p(*med, d, light_dir));
                    else direct_spec = surface_eval<BM>(bx, it, d, light_dir, sc.world_ior, p.two_sides);
                    float mis_w = 1.0f;
                    if (p.use_mis && !(src.bool_bits & 0x01)) {
                        const float light_pdf = emitter_pdf * direct_pdf;
                        const float bsdf_pdf_v = MI ? direct_spec.x : surface_pdf<BM>(bx, it, light_dir, d, sc.world_ior, p.two_sides);
                        mis_w = balance(light_pdf, bsdf_pdf_v);
                    }
                    if (isnan(mis_w)) { stL(q.L, p.cap, l_off, splat3(mis_w)); poisoned = true; }     // as in k_shade: the sample is zeroed at the end
                    else {
                        f3 c = (direct_spec * shadow_int) * mis_w;
                        if (ns != 1) c = c / emitter_pdf;
                        contrib = (c * p.inv_S) * thr;
                        want = !(contrib.x == 0.f && contrib.y == 0.f && contrib.z == 0.f);
                    }
                }
            }
            t_shadow += wave_count(sampled); t_poison += wave_count(poisoned);
            const uint32_t spos = wave_append(want, shadow_counter);
            if (want && spos < q.sh_subcap) {
                const uint32_t so = (sh_qbase + spos) << 2, sc_ = q.sh_cap;
                st3q(q.sh_o, sc_, so, hit_point);
                st3q(q.sh_d, sc_, so, light_dir);
                stq(q.sh_tmax, so, emitter_d);
                st3q(q.sh_c, sc_, so, contrib);
                stq(q.sh_id, so, l_off | ((p.l_planes > 1) ? (uint32_t)s : 0u));
            }
        }

        // ---- Steps 5-6: emission of the surface we are on, the next direction, the emission weight of the NEXT hit
        f3 new_d = d;
        bool is_spec = false, cont = false;
        if (shade) {
            if (!MI && (SM & 2) && hit_light >= 0) {
                const f3 emit_int = emitter_eval_le(sc.src[hit_light], hit_point - o, it.n_g);      // geometric normal here (vpt.py:233)
                if (!(emit_int.x == 0.f && emit_int.y == 0.f && emit_int.z == 0.f)) {
                    const f3 add = (emit_int * emission_weight) * thr;
                    add_radiance(q.L, p.cap, l_off, add, true);
                }
            }
            float ray_pdf = 1.f;
            if (MI) {                                                           // Medium.sample_new_rays, medium.py:112-121
                if (VOL && vol_event) {                                         // GridVolume.sample_new_rays: its own phase function
                    const f3 local = phase_sample_p(sc.vol.ph, d, rng, ray_pdf);
                    new_d = delocalize(d, local);
                } else if (med->type >= 0) {
                    const f3 local = phase_sample_p(*med, d, rng, ray_pdf);
                    new_d = delocalize(d, local);
                }
                cont = true;                                                    // a medium event never ends the path by itself
            } else {
                f3 spec;
                new_d = surface_sample<BM>(bx, it, d, sc.world_ior, p.two_sides, rng, spec, ray_pdf, is_spec);
                cont = !(max3(spec) == 0.f || ray_pdf == 0.f);                  // vpt.py:240-241
                if (cont) thr = thr * (spec / ray_pdf);
            }
            bounce += 1;
            if ((int)bounce >= p.max_bounce) cont = false;
            if (cont && it.obj_id >= 0) {                                       // vpt.py:247-253, with THIS interaction
                hit_light = rec_light;
                if (p.use_mis) {
                    float e_pdf = 0.0f;
                    if (hit_light >= 0 && bx.is_delta == 0 && !is_spec) e_pdf = emitter_solid_angle_pdf(sc.src[hit_light], it, new_d);
                    emission_weight = balance(ray_pdf, e_pdf);
                }
            }
        }
        if (rng.draw != draw0) atomicAdd(&s_draws[threadIdx.x >> 6], rng.draw - draw0);
        const uint32_t npos = wave_append(cont, next_counter);
        if (cont) {
            const uint32_t so = (qbase + npos) << 2;
            st3q(q.ray_o[nxt], p.cap, so, hit_point);
            st3q(q.ray_d[nxt], p.cap, so, new_d);
            st3q(q.thr[nxt], p.cap, so, thr);
            stq(q.id[nxt], so, id);
            stq(q.meta[nxt], so, (rng.draw & 0x7fffffu) | (bounce << 23));      // tracking loops draw thousands of numbers per path: 16 bits would wrap
            if (rng.draw > 0x7fffffu) atomicAdd(&cnt->stats[sl.q][ST_OVERFLOW], 1ull);
            stq(q.pdf[nxt], so, emission_weight);
        }
    }
    flush_uniform(t_shade, &cnt->stats[sl.q][ST_SHADE]);
    flush_uniform(t_shadow, &cnt->stats[sl.q][ST_SHADOW]);
    if (lane_id() == 0 && s_draws[threadIdx.x >> 6]) atomicAdd(&cnt->stats[sl.q][ST_DRAWS], (unsigned long long)s_draws[threadIdx.x >> 6]);
    flush_uniform(t_poison, &cnt->stats[sl.q][ST_POISON]);
}
template <int BM, int SM, int VOL, int MI>
__global__ void __launch_bounds__(BLOCK) k_vshade_ev(DevScene sc, Params p, Queues q, Counters* cnt, int cls, int cur) {
    vshade_ev_body<BM, SM, VOL, MI>(sc, p, q, cnt, cls, cur);
}
// ---- event kernels in groups: ONE launch shades several event queues, as k_shade_group does for the surface tracer's class queues (the
// launch boundary between two short kernels costs more than either: V2 ran five event kernels of 24-57 us per iteration).  A kernel
// allocates for its largest member, so the groups follow the footprints (api.hip kVGroup): four waves per SIMD (Lambertian, Oren-Nayar,
// delta, Lambertian transmission: 103-124 VGPRs), three (the MEDIUM kernel, thin coat, microfacet: 120-156), two (Blinn-Phong, modified
// Phong, Fresnel blend: 169-210).  A member code is a material mask, or APT_VEV_MEDIUM_CODE for the medium kernel; queues absent from
// the scene are skipped by a wave-uniform test.  Same code per event, same order inside a queue: images and statistics are those of one
// launch per queue.
#define APT_VEV_MEDIUM_CODE 0x10000
struct VGroupIn { int cls[4]; };          // per member: its event queue (-1: not in this scene)
template <int B, int SM, int VOL>
APT_D void vshade_ev_member(const DevScene& sc, const Params& p, const Queues& q, Counters* cnt, int cls, int cur) {
    if constexpr (B == APT_VEV_MEDIUM_CODE) vshade_ev_body<0x000, SM, VOL, 1>(sc, p, q, cnt, cls, cur);
    else vshade_ev_body<B, SM, VOL, 0>(sc, p, q, cnt, cls, cur);
}
template <int SM, int VOL, int WAVES, int B0, int B1, int B2, int B3>
__global__ void __launch_bounds__(BLOCK, WAVES) k_vshade_ev_group(DevScene sc, Params p, Queues q, Counters* cnt, VGroupIn g, int cur) {
    // (scene, parameters and queues are read through the kernel-argument segment, the pointer made opaque per member: left to itself the
    // compiler loads the fields of all four members up front - 240 scalar registers spilled, some of them to scratch; shade_stage.hpp)
    const args3_ptr A0 = kernel_args3();
    if constexpr (B0 != 0) if (g.cls[0] >= 0) { const ShadeArgs3* A = args_fresh(A0); vshade_ev_member<B0, SM, VOL>(A->sc, A->p, A->q, cnt, g.cls[0], cur); }
    if constexpr (B1 != 0) if (g.cls[1] >= 0) { const ShadeArgs3* A = args_fresh(A0); vshade_ev_member<B1, SM, VOL>(A->sc, A->p, A->q, cnt, g.cls[1], cur); }
    if constexpr (B2 != 0) if (g.cls[2] >= 0) { const ShadeArgs3* A = args_fresh(A0); vshade_ev_member<B2, SM, VOL>(A->sc, A->p, A->q, cnt, g.cls[2], cur); }
    if constexpr (B3 != 0) if (g.cls[3] >= 0) { const ShadeArgs3* A = args_fresh(A0); vshade_ev_member<B3, SM, VOL>(A->sc, A->p, A->q, cnt, g.cls[3], cur); }
}

// ------------------------------------------------------------------ vshadow
// track_ray (vpt.py:99-138), one closest-hit query per pass.  Pass 0 reads every queued light sample; a sample that is blocked by
// a non-null surface is dropped, one that reaches its light adds contribution x transmittance to the path's radiance, and one
// that crossed a null surface (or a stretch of scattering world) is written back in place - origin moved to the crossing point,
// remaining distance shortened, transmittance folded into the contribution - and its slot is appended to the list the next pass
// reads.  The reference walks at most seven segments; the host launches pass p + 1 only where null surfaces exist.
template <int MODE>
__global__ void __launch_bounds__(VSHADOW_NT(MODE), (MODE == 2 ? APT_VSHADOW_WAVES : 1)) k_vshadow(DevScene sc, Params p, Queues q, Counters* cnt, LdsPlan plan, int pass) {
    __shared__ float s_sweep[MODE == 1 ? APT_SWEEP_LDS_FLOATS(BLOCK) : 1];
    const SubLoop sl = sub_loop(p.nq, VSHADOW_NT(MODE));
    const uint32_t n = min(pass == 0 ? cnt->n_shadow[sl.q * CNT_PAD] : cnt->n_walk[pass][sl.q * CNT_PAD], q.sh_subcap);
    if (pass == 0 && sl.first == 0 && threadIdx.x == 0) {
        cnt->stats[sl.q][ST_SHADOW_TRACED] += n;
        for (int c = 0; c < q.n_classes; c++) cnt->n_cls[c][sl.q * CNT_PAD] = 0;      // every shade of this iteration is done
    }
    const uint32_t qbase = (uint32_t)sl.q * q.sh_subcap, sc_ = q.sh_cap;
    const uint32_t* list_in = q.sh_walk[pass & 1];
    uint32_t* list_out = q.sh_walk[(pass + 1) & 1];
    uint32_t* next_counter = &cnt->n_walk[pass + 1][sl.q * CNT_PAD];
    const f3 world_ue = sc.med[sc.n_objects].u_e;
    const bool world_scat = sc.med[sc.n_objects].type >= 0;
    uint32_t t_lit = 0, t_track = 0;
    for (uint32_t base = sl.first; base < n; base += sl.stride) {
        const uint32_t pos = base + threadIdx.x;
        const bool valid = pos < n;
        uint32_t idx = qbase + (valid ? pos : n - 1);
        if (pass > 0) idx = ldq(list_in, idx << 2);                         // slot of a sample that is still walking
        const uint32_t io = idx << 2;
        f3 o = ld3q(q.sh_o, sc_, io);
        const f3 d = ld3q(q.sh_d, sc_, io);
        float depth = ldq(q.sh_tmax, io);
        HitRec rec; rec.t = (depth > 0.0f) ? depth - 1e-4f : 1e7f; rec.prim = -1; rec.u = rec.v = 0.f;
        if (MODE == 0) traverse<false>(sc.bvh, make_stack(plan), o, d, rec);
        else if (MODE == 1) sweep_wg<false, BLOCK>(sc.sweep, o, d, rec, valid, s_sweep);
        else sweep_tile<false, APT_VSHADOW_NT>(sc.sweep, o, d, rec, valid, reinterpret_cast<float*>(s_dyn));
        bool arrived = false, walk_on = false;
        f3 c = splat3(0.f);
        if (valid) {
            t_track++;
            c = ld3q(q.sh_c, sc_, io);
            int obj = -1; bool in_free = true, blocked = false; float seg = depth;
            if (rec.prim < 0) { if (!world_scat) arrived = true; }         // nothing in the way and nothing to attenuate: done
            else {
                obj = sc.prim_obj[rec.prim];
                if (vpt_non_null(sc, obj)) blocked = true;
                else {
                    Hit it; build_hit(sc, rec.prim, rec.t, rec.u, rec.v, o, d, it);
                    in_free = dot(it.n_g, d) < 0.f;
                    seg = rec.t;
                }
            }
            if (blocked) c = c * 0.f;                                       // NaN for a non-finite c, exactly as upstream's 0 * x
            else if (!arrived) {
                // get_transmittance, vpt.py:52-62
                if (in_free && world_scat) c = c * exp_neg(world_ue, seg);
                else if (!in_free && vpt_is_scattering(sc, obj)) c = c * exp_neg(sc.med[obj].u_e, seg);
                o = o + d * seg;
                depth -= seg;
                if (depth <= 5e-5f || pass >= 6) arrived = true;            // at the light, or the seventh segment (vpt.py:113)
                else walk_on = true;
            }
            if (arrived || blocked) {
                if (!(c.x == 0.f && c.y == 0.f && c.z == 0.f)) {
                    add_radiance(q.L, p.cap, ldq(q.sh_id, io), c, APT_EXCLUSIVE_L(p));      // (one radiance plane per light sample of a vertex, or one sample: no two entries of a launch share a slot)
                }
                if (arrived) t_lit++;
            }
        }
        if (list_out != nullptr) {
            const uint32_t wpos = wave_append(walk_on, next_counter);
            if (walk_on && wpos < q.sh_subcap) {
                st3q(q.sh_o, sc_, io, o); stq(q.sh_tmax, io, depth); st3q(q.sh_c, sc_, io, c);
                stq(list_out, (qbase + wpos) << 2, idx);
            }
        }
    }
    flush_stat(t_lit, &cnt->stats[sl.q][ST_LIT]);
    flush_stat(t_track, &cnt->stats[sl.q][ST_TRACK]);
}

#ifndef APT_VSHADOW_FLAT_PASSES
#define APT_VSHADOW_FLAT_PASSES 3      // launches per iteration in scenes with null surfaces: segment 0, segment 1, the rest
#endif
#if APT_FAST
// track_ray on the flat sweep (product build, scenes of up to APT_FLAT_MAX_PRIMS primitives): TWO light samples per lane - the two halves of
// every packed instruction of the sweep (traverse.hpp flat_closest2).  k_vshadow<tile> tests a lane's one ray with the reference's
// arithmetic through workgroup lists in LDS; this kernel keeps its pass structure (pass 0 reads the queue, pass p the slot list of the
// samples that crossed a null surface in pass p - 1: a sweep is wave-uniform over the records, so the survivors have to be packed densely
// again or later segments cost what the first one did - measured with all segments in one launch: V1 walk 38.5 -> 33.0 ms only) and
// replaces the intersector.  Per sample the arithmetic after the hit is k_vshadow's; the hit itself is the flat sweep's (t
// within 1e-5 relative, same primitive unless tied; coplanar near-ties and zero-component directions settled by the reference-order code
// inside flat_closest2).
__global__ void __launch_bounds__(BLOCK) k_vshadow_flat(DevScene sc, Params p, Queues q, Counters* cnt, LdsPlan plan, int pass) {
    const SubLoop sl = sub_loop(p.nq, FLAT_NT);
    const uint32_t n = min(pass == 0 ? cnt->n_shadow[sl.q * CNT_PAD] : cnt->n_walk[pass][sl.q * CNT_PAD], q.sh_subcap);
    if (pass == 0 && sl.first == 0 && threadIdx.x == 0) {
        cnt->stats[sl.q][ST_SHADOW_TRACED] += n;
        for (int c = 0; c < q.n_classes; c++) cnt->n_cls[c][sl.q * CNT_PAD] = 0;      // every shade of this iteration is done
    }
    const uint32_t qbase = (uint32_t)sl.q * q.sh_subcap, sc_ = q.sh_cap;
    const uint32_t* list_in = q.sh_walk[pass & 1];
    uint32_t* list_out = q.sh_walk[(pass + 1) & 1];
    uint32_t* next_counter = &cnt->n_walk[pass + 1][sl.q * CNT_PAD];
    const f3 world_ue = sc.med[sc.n_objects].u_e;
    const bool world_scat = sc.med[sc.n_objects].type >= 0;
    const bool excl = APT_EXCLUSIVE_L(p);
    uint32_t t_lit = 0, t_track = 0;
    // one segment of one sample, given its closest hit (k_vshadow's per-sample block); returns true when the sample has to walk on
    auto segment = [&](bool valid, f3& o, const f3& d, float& depth, f3& c, const HitRec& rec, uint32_t io, int seg_no) -> bool {
        if (!valid) return false;
        t_track++;
        int obj = -1; bool in_free = true, blocked = false, arrived = false, walk_on = false; float seg = depth;
        if (rec.prim < 0) { if (!world_scat) arrived = true; }             // nothing in the way and nothing to attenuate: done
        else {
            obj = sc.prim_obj[rec.prim];
            if (vpt_non_null(sc, obj)) blocked = true;
            else {
                Hit it; build_hit(sc, rec.prim, rec.t, rec.u, rec.v, o, d, it);
                in_free = dot(it.n_g, d) < 0.f;
                seg = rec.t;
            }
        }
        if (blocked) c = c * 0.f;                                           // NaN for a non-finite c, exactly as upstream's 0 * x
        else if (!arrived) {
            // get_transmittance, vpt.py:52-62
            if (in_free && world_scat) c = c * exp_neg(world_ue, seg);
            else if (!in_free && vpt_is_scattering(sc, obj)) c = c * exp_neg(sc.med[obj].u_e, seg);
            o = o + d * seg;
            depth -= seg;
            if (depth <= 5e-5f || seg_no >= 6) arrived = true;              // at the light, or the seventh segment (vpt.py:113)
            else walk_on = true;
        }
        if (arrived || blocked) {
            if (!(c.x == 0.f && c.y == 0.f && c.z == 0.f)) add_radiance(q.L, p.cap, ldq(q.sh_id, io), c, excl);
            if (arrived) t_lit++;
        }
        return walk_on;
    };
    for (uint32_t base = sl.first; base < n; base += sl.stride) {
        const uint32_t pos = base + 2u * threadIdx.x;
        const bool v0 = pos < n, v1 = pos + 1u < n;
        uint32_t i0 = qbase + (v0 ? pos : n - 1u), i1 = qbase + (v1 ? pos + 1u : n - 1u);
        if (pass > 0) { i0 = ldq(list_in, i0 << 2); i1 = ldq(list_in, i1 << 2); }      // slots of samples that are still walking
        const uint32_t io0 = i0 << 2, io1 = i1 << 2;
        f3 o0 = ld3q(q.sh_o, sc_, io0), o1 = ld3q(q.sh_o, sc_, io1);
        const f3 d0 = ld3q(q.sh_d, sc_, io0), d1 = ld3q(q.sh_d, sc_, io1);
        float depth0 = ldq(q.sh_tmax, io0), depth1 = ldq(q.sh_tmax, io1);
        f3 c0 = ld3q(q.sh_c, sc_, io0), c1 = ld3q(q.sh_c, sc_, io1);
        bool on0 = v0, on1 = v1;
        for (int seg = pass; ; seg++) {
            HitRec r0, r1; int k0, k1;
            r0.t = !on0 ? -1.0f : ((depth0 > 0.0f) ? depth0 - 1e-4f : 1e7f); r0.prim = -1; r0.u = r0.v = 0.f;      // (a finished sample searches below a negative limit: nothing is accepted)
            r1.t = !on1 ? -1.0f : ((depth1 > 0.0f) ? depth1 - 1e-4f : 1e7f); r1.prim = -1; r1.u = r1.v = 0.f;
            flat_closest2(sc.flat, sc.sweep, sc.prim_class, o0, d0, o1, d1, r0, r1, k0, k1);
            on0 = segment(on0, o0, d0, depth0, c0, r0, io0, seg);
            on1 = segment(on1, o1, d1, depth1, c1, r1, io1, seg);
            // the last launch of an iteration (pass APT_VSHADOW_FLAT_PASSES - 1) walks what is left of its samples' segments itself: by then a
            // handful of samples are still going (V1: a tenth after two segments), and four more launches over near-empty lists cost more than a thin wave
            if (pass < APT_VSHADOW_FLAT_PASSES - 1 || seg >= 6 || !__any(on0 || on1)) break;
        }
        if (list_out != nullptr && pass < APT_VSHADOW_FLAT_PASSES - 1) {
            // both entries of the lane with ONE tail atomic per wave: the wave's first entries, then its second ones
            const unsigned long long m0 = __ballot(on0), m1 = __ballot(on1);
            uint32_t tail = 0;
            if (lane_id() == 0 && (m0 | m1)) tail = atomicAdd(next_counter, (uint32_t)(__popcll(m0) + __popcll(m1)));
            tail = (uint32_t)__builtin_amdgcn_readlane((int)tail, 0);
            const uint32_t w0 = tail + rank_in(m0), w1 = tail + (uint32_t)__popcll(m0) + rank_in(m1);
            if (on0 && w0 < q.sh_subcap) { st3q(q.sh_o, sc_, io0, o0); stq(q.sh_tmax, io0, depth0); st3q(q.sh_c, sc_, io0, c0); stq(list_out, (qbase + w0) << 2, i0); }
            if (on1 && w1 < q.sh_subcap) { st3q(q.sh_o, sc_, io1, o1); stq(q.sh_tmax, io1, depth1); st3q(q.sh_c, sc_, io1, c1); stq(list_out, (qbase + w1) << 2, i1); }
        }
    }
    flush_stat(t_lit, &cnt->stats[sl.q][ST_LIT]);
    flush_stat(t_track, &cnt->stats[sl.q][ST_TRACK]);
}
#endif

// ------------------------------------------------------- unit entry kernel
// Medium functions on explicit inputs (parity probe): one medium row and 7 input floats per test, 8 output floats; RNG = Philox
// stream keyed by (test index, seed, 1).  mode 0 sample_mfp, 1 sample_new_rays, 2 phase value + transmittance.
__global__ void k_medium_probe(int n, const DevMedium* med, int mode, const float* in7, uint32_t seed, float* out8) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const DevMedium m = med[k];
    const float* x = in7 + 7 * k; float* y = out8 + 8 * k;
    Philox r; rng_init(r, (uint32_t)k, seed, 1u, 0u);
    if (mode == 0) {
        float t; f3 beta; const bool is_mi = medium_sample_mfp(m, x[0], r, t, beta);
        y[0] = is_mi ? 1.f : 0.f; y[1] = t; y[2] = beta.x; y[3] = beta.y; y[4] = beta.z; y[5] = (float)r.draw;
    } else if (mode == 1) {
        const f3 incid = ld3(x); f3 d = incid; float p = 1.f;
        if (m.type >= 0) { const f3 local = phase_sample_p(m, incid, r, p); d = delocalize(incid, local); }
        y[0] = d.x; y[1] = d.y; y[2] = d.z; y[3] = p; y[4] = p; y[5] = p; y[6] = p; y[7] = (float)r.draw;
    } else {
        y[0] = (m.type >= 0) ? phase_eval_p(m, ld3(x), ld3(x + 3)) : 1.f;
        const f3 tr = exp_neg(m.u_e, x[6]);
        y[1] = tr.x; y[2] = tr.y; y[3] = tr.z;
    }
}
