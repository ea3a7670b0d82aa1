// shade_stage.hpp - the shade stage of the wavefront path tracer (vanilla_renderer.py:36-117): what a path does at a vertex between two
// closest-hit queries - emission and its MIS weight, Russian roulette, one shadow ray per useful light sample, the continuation ray.
//
// Two kernels are made of the same three shading steps (open_vertex, sample_light, emit_and_scatter):
//   k_shade / k_shade_group  the STAGED kernel: reads what an extend kernel wrote (SoA queues, or the packed queue of one material
//                            class) and writes the next ray queue and the shadow queue (both builds, every trace mode);
//   k_shade_traced           the kernel that TRACES ITS OWN RAYS (product build, flat sweep, unsorted, one light sample per vertex -
//                            C1 / C2): continuation ray and light sample are swept against the scene's records in place, a bounce is
//                            one launch, and the path's radiance travels with its record.
#pragma once
#include "stages.hpp"

// ----------------------------------------------------------------- textures
// Taichi's float `a % b` is a - b * floor(a / b) (python/taichi/lang/ops.py, mod)
APT_D float ti_fmod(float a, float b) { float q = floorf(a / b); return a - b * q; }
APT_D f3 mix3(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }          // taichi.math.mix: x * (1 - a) + y * a
// Texture.query, bxdf/texture.py:111-139: bilinear lookup inside the texture's rectangle of the atlas
APT_D f3 texture_query(const DevScene& sc, int map, int obj, float u, float v) {
    const int* ti_ = sc.tex_i + 15 * obj + 5 * map; const float* tf = sc.tex_f + 6 * obj + 2 * map;
    const float w = (float)ti_[3], h = (float)ti_[4];
    const float scaled_u = ti_fmod((u * tf[0]) * w, w - 1.f), scaled_v = ti_fmod((v * tf[1]) * h, h - 1.f);
    float floor_u = floorf(scaled_u), floor_v = floorf(scaled_v);
    const float ratio_u = scaled_u - floor_u, ratio_v = scaled_v - floor_v;
    floor_u = floor_u + (float)ti_[1]; floor_v = floor_v + (float)ti_[2];
    const int fu = (int)floor_u, fv = (int)floor_v;
    const float* img = sc.atlas[map]; const int W = sc.atlas_w[map];
    const float* r0 = img + ((size_t)fv * W + fu) * 3; const float* r1 = r0 + (size_t)W * 3;
    const f3 q_ff = mk3(r0[0], r0[1], r0[2]), q_cf = mk3(r0[3], r0[4], r0[5]), q_fc = mk3(r1[0], r1[1], r1[2]), q_cc = mk3(r1[3], r1[4], r1[5]);
    return mix3(mix3(q_ff, q_cf, ratio_u), mix3(q_fc, q_cc, ratio_u), ratio_v);
}
// PathTracer.get_uv_item, path_tracer.py:276-289 (meshes only: textured spheres are refused at scene creation)
APT_D bool get_uv_item(const DevScene& sc, int map, int obj, int prim, float bu, float bv, f3& out) {
    if (sc.atlas[map] == nullptr || !(sc.tex_i[15 * obj + 5 * map] > -255)) return false;
    const float* uv = sc.uvs + 6 * prim;
    const float w0 = 1.f - bu - bv;
    const float gu = (uv[2] * bu + uv[4] * bv) + uv[0] * w0, gv = (uv[3] * bu + uv[5] * bv) + uv[1] * w0;
    out = texture_query(sc, map, obj, gu, gv);
    return true;
}

// -------------------------------------------------------------------- shade
APT_D void build_hit_rec(const DevScene& sc, float4 ra, float4 rb, int prim, float t, float u, float v, f3 o, f3 d, Hit& it, int& hit_light, f3& k_d, bool with_vn = true);
APT_D void build_hit(const DevScene& sc, int prim, float t, float u, float v, f3 o, f3 d, Hit& it, int& hit_light, f3& k_d) {
    build_hit_rec(sc, sc.prim_shade[2 * prim], sc.prim_shade[2 * prim + 1], prim, t, u, v, o, d, it, hit_light, k_d);
}
APT_D void build_hit_rec(const DevScene& sc, float4 ra, float4 rb, int prim, float t, float u, float v, f3 o, f3 d, Hit& it, int& hit_light, f3& k_d, bool with_vn) {
    const int code = __float_as_int(ra.w);
    it.prim_id = prim; it.min_depth = t;
    it.obj_id = (code < 0) ? ~code : code;
    hit_light = __float_as_int(rb.x);
    k_d = mk3(rb.y, rb.z, rb.w);
    if (code < 0) {
        // sphere: the record holds the centre; normal from the hit point (tracer_base.py:217-223)
        it.n_g = normalize((o + d * t) - mk3(ra.x, ra.y, ra.z));
        it.n_s = it.n_g;
    } else {
        it.n_g = mk3(ra.x, ra.y, ra.z);
        if (with_vn && sc.has_vn) {
            const float4 v0 = sc.vnormals[3 * prim], v1 = sc.vnormals[3 * prim + 1], v2 = sc.vnormals[3 * prim + 2];
            // interpolated vertex normal, NOT re-normalised (tracer_base.py:228-230)
            it.n_s = (mk3(v0.x, v0.y, v0.z) * (1.f - u - v) + mk3(v1.x, v1.y, v1.z) * u) + mk3(v2.x, v2.y, v2.z) * v;
        } else it.n_s = it.n_g;
    }
}
APT_D void build_hit(const DevScene& sc, int prim, float t, float u, float v, f3 o, f3 d, Hit& it) {
    int light; f3 kd; build_hit(sc, prim, t, u, v, o, d, it, light, kd);
}

#if APT_FAST
// ---- rays traced in place (Params::fused == 2; product build, flat sweep, unsorted, one light sample per vertex)
// The shade kernel sweeps its continuation ray against the scene's records itself (flat_closest1: one ray against two records per packed
// instruction), as it already does with its light sample, and k_generate does the same for the camera rays: a bounce is ONE launch
// instead of extend + fix-up + shade, the ray is never read back (24 + 8 bytes per segment), and a ray that hits nothing never enters a
// queue - its path ends where it was sampled (18 % of C2's continuation rays: no record written, no idle lane in the next launch).
// The rare rays that need the reference's own arithmetic (traverse.hpp flat_closest2: near-tied coplanar faces, directions for which
// upstream's slab cull is part of the result) are queued with a provisional record and listed, as before - but the lists are served by
// the NEXT launch itself instead of a fix-up launch per bounce (a launch boundary is a pipeline drain: ~20 us of a render lane each):
// a wave that finds its sub-queue's lists non-empty claims them (one atomic), serves them - one entry per lane, the full reference-order
// code - and publishes "done"; the sub-queue's other waves wait for that before they read a record.  The lists are empty in all but a few
// launches per render, where the whole protocol is two scalar loads per wave; the serving code sits in front of the kernel's main loop,
// where almost no register is live, so it costs the hot loop nothing (k_fix_flat alone allocates 84 VGPRs, the shade kernel 122).
// Nothing depends on how workgroups are placed: whichever wave claims a list is running, hence the waiters cannot starve it.
APT_D void fix_prologue(const DevScene& sc, const Params& p, const Queues& q, Counters* cnt, int cur, int sq, int bounce) {
    const uint32_t epoch = (uint32_t)bounce + 1u;
    const int ncls = q.tr_ncls;
    uint32_t* n_def_p = &cnt->n_tr[bounce % 3][ncls][sq * CNT_PAD]; uint32_t* n_sh_p = &cnt->n_fix_sh[cur ^ 1][sq * CNT_PAD];
    // Memory order.  The list lengths are read first (the claim word below is read AFTER them), then the claim word.  A wave that
    // finds this bounce claimed - by a running or a finished server - waits for "done" with an acquire load whatever lengths it saw, so
    // that everything the server appended or added (queue records, counters, radiance slots) happens-before this wave's reads; the
    // server resets the light-sample list BEFORE it publishes "done", and only after its claim, so lengths of zero seen together with an
    // unclaimed bounce are the lists' true lengths.
    // (relaxed loads performed at L2 and a WORKGROUP-scope fence - a wait for the loads, no cache invalidate - give the load-load order;
    // acquire loads at agent scope put a `buffer_inv` behind each of them in every wave of every launch: C1 4 709 -> 3 442 Msamples/s, C2's
    // shade kernel 18.2 -> 20.0 ms per 256 spp, measured.  The acquire that matters is the one on "done" below.)
    const uint32_t n_def = __hip_atomic_load(n_def_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), n_sh = min(__hip_atomic_load(n_sh_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), q.sh_subcap);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t old = __hip_atomic_load(&cnt->fix_claim[sq * CNT_PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old < epoch) {
        if ((n_def | n_sh) == 0u) return;                     // nothing listed, nobody serving
        if (lane_id() == 0) old = atomicMax(&cnt->fix_claim[sq * CNT_PAD], epoch);
        old = (uint32_t)__builtin_amdgcn_readlane((int)old, 0);
    }
    if (old >= epoch) {                                       // somebody else serves (or has served) the lists of this bounce
        while (__hip_atomic_load(&cnt->fix_done[sq * CNT_PAD], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(16);
        return;
    }
    const uint32_t qbase = (uint32_t)sq * p.subcap, sh_qbase = (uint32_t)sq * q.sh_subcap;
    const bool need_uv = sc.has_vn || sc.tex_i != nullptr;
    for (uint32_t base = 0; base < n_def; base += 64u) {      // staged rays: closest hit by the reference-order code, then the record joins its class queue (or the path ends)
        const uint32_t li = base + lane_id(); const bool valid = li < n_def;
        // (staging at the top of the sub-queue's own region: served from its LOWEST slot upwards - a resolved record is appended at or below the
        // slot its ray was just read from, never onto a staged ray that is still to be served)
        const uint32_t lj = valid ? li : n_def - 1u;
        const uint32_t io = (q.tr_stage_top ? qbase + p.subcap - n_def + lj : (uint32_t)ncls * p.cap + qbase + lj) << 4;
        const float4 ra = ldq(q.tr[cur][0], io), rb = ldq(q.tr[cur][1], io), rc = ldq(q.tr[cur][2], io), rd = ldq(q.tr[cur][3], io);
        const f3 o = mk3(ra.x, ra.y, ra.z), d = mk3(rb.x, rb.y, rb.z);
        HitRec r0, r1; r0.t = r1.t = 1e7f; r0.prim = r1.prim = -1; r0.u = r0.v = r1.u = r1.v = 0.f;
        int c0, c1;
        flat_closest2(sc.flat, sc.sweep, sc.prim_class, o, d, o, d, r0, r1, c0, c1);
        if (valid && r0.prim >= 0) {
            const int oc = ncls > 1 ? c0 : 0;
            const uint32_t pos = atomicAdd(&cnt->n_tr[bounce % 3][oc][sq * CNT_PAD], 1u);      // (one atomic per entry: this path is a handful of rays per million)
            const uint32_t slot = (uint32_t)oc * p.cap + qbase + pos, so = slot << 4;
            stq(q.tr[cur][0], so, make_float4(ra.x, ra.y, ra.z, r0.t));
            stq(q.tr[cur][1], so, make_float4(rb.x, rb.y, rb.z, __uint_as_float((__float_as_uint(rb.w) & ~0xffu) | (uint32_t)r0.prim)));
            stq(q.tr[cur][2], so, rc); stq(q.tr[cur][3], so, rd);
            if (need_uv) { float2 uv_; uv_.x = r0.u; uv_.y = r0.v; stq(q.tr_uv[cur], slot << 3, uv_); }
        } else if (valid && !(rd.x == 0.f && rd.y == 0.f && rd.z == 0.f)) {      // nothing hit: the path ends, its radiance goes to its slot
            const uint32_t id = __float_as_uint(rc.w), lp_ = id & ((1u << p.pix_bits) - 1u), s_ = id >> p.pix_bits;
            add_radiance(q.L, p.cap, (s_ * (uint32_t)p.npix + lp_) << 2, mk3(rd.x, rd.y, rd.z), true);
        }
    }
    uint32_t t_lit = 0;
    for (uint32_t base = 0; base < n_sh; base += 64u) {       // light samples the previous bounce could not settle (shadow_flat_body<3>)
        const uint32_t li = base + lane_id(); const bool valid = li < n_sh;
        const uint32_t io = (sh_qbase + (valid ? li : n_sh - 1u)) << 2;
        const f3 o = ld3q(q.sh_o, q.sh_cap, io), d = ld3q(q.sh_d, q.sh_cap, io), c = ld3q(q.sh_c, q.sh_cap, io);
        const float dist = ldq(q.sh_tmax, io); const uint32_t slot = ldq(q.sh_id, io);
        bool occ, occ_b;
        const float lim = (dist > 0.0f) ? dist - 1e-4f : 1e7f;
        flat_any2(sc.flat, sc.sweep, o, d, o, d, lim, lim, occ, occ_b);
        // (several samples of one vertex may be listed - S > 1 - and share its slot: one entry at a time within the wave's 64)
        const bool weird = !(isfinite(c.x) && isfinite(c.y) && isfinite(c.z));
        const bool add = valid && (!occ || weird);
        for (unsigned long long m = __ballot(add); m != 0ull; m &= m - 1ull) {
            if ((int)lane_id() == __ffsll((long long)m) - 1) add_radiance(q.L, p.cap, slot, occ ? c * 0.f : c, true);      // (k_shadow: an occluded non-finite sample enters upstream's sum as 0 * contribution)
        }
        t_lit += (valid && !occ) ? 1u : 0u;
    }
    flush_stat(t_lit, &cnt->stats[sq][ST_LIT]);
    if (lane_id() == 0) __hip_atomic_store(n_sh_p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // consumed - before "done" is published (the bounce after this one appends to this list again; the staging queue's counter rotates with the others)
    __threadfence();
    if (lane_id() == 0) __hip_atomic_store(&cnt->fix_done[sq * CNT_PAD], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
// The shade kernels' scalar operands.  Scene, parameters and queues arrive by value in the kernel-argument segment - ~1.8 KB, of which a
// tile row touches ~165 dwords - and left alone the compiler loads every field it needs ONCE, in front of the row loop, where 102 scalar
// registers cannot hold them: the rest lives in lanes of two spill VGPRs and comes back through v_readlane at every use (242 of the 2 200
// VALU instructions of C2's row loop - on the unit this kernel is bound by).  So the arguments are read THROUGH THE SEGMENT POINTER, and
// the pointer is made opaque at the head of every phase of a row (an empty asm: APT_ARGS_PHASE): a field's load can then not be hoisted
// above the phase that uses it, it becomes an s_load (scalar memory, not a VALU issue slot; the segment stays in the scalar cache) next
// to its use, and its register is free again after the phase.
#ifndef APT_ARGS_RELOAD
#define APT_ARGS_RELOAD 1
#endif
struct ShadeArgs3 { DevScene sc; Params p; Queues q; };          // the leading arguments of every shade kernel, laid out as the segment lays them out
typedef const __attribute__((address_space(4))) ShadeArgs3* args3_ptr;
APT_D args3_ptr kernel_args3() { return (args3_ptr)__builtin_amdgcn_kernarg_segment_ptr(); }
APT_D const ShadeArgs3* args_fresh(args3_ptr a) {
#if APT_ARGS_RELOAD
    asm volatile("" : "+s"(a));
#endif
    return (const ShadeArgs3*)a;
}
#if APT_ARGS_RELOAD
#define APT_ARGS_PHASE() (A_ = args_fresh(A0))
#else
#define APT_ARGS_PHASE() ((void)0)
#endif
// ---- a vertex while it is shaded, and the three steps both shade kernels take with it (vanilla_renderer.py:36-120)
struct Vertex {
    f3 o, d, thr, hit_point; uint32_t id, l_off, draw0; float emission_weight;      // l_off: byte offset of the path's radiance slot; draw0: the path's draw index on entry
    Hit it; int hit_light; DevBxdf bx;
};
APT_D void vertex_reset(Vertex& vx) {
    vx.o = splat3(0.f); vx.d = mk3(0.f, 0.f, 1.f); vx.thr = splat3(0.f); vx.hit_point = splat3(0.f); vx.id = 0; vx.l_off = 0; vx.draw0 = 0; vx.emission_weight = 1.0f;
    vx.it.obj_id = -1; vx.it.prim_id = -1; vx.it.n_s = vx.it.n_g = mk3(1.f, 0.f, 0.f); vx.it.min_depth = 0.f; vx.hit_light = -1;
    vx.bx.type = 1; vx.bx.is_delta = 0; vx.bx.is_bsdf = 0; vx.bx.k_d = vx.bx.k_s = vx.bx.k_g = vx.bx.mean = splat3(0.f); vx.bx.ior = 1.f;
}
// Step 1, after the hit has been built (vx.it, vx.hit_light; rec_kd: the colour in the primitive's record): material and textures, the
// path's radiance slot and random stream, the emission MIS weight of this hit (the tail of the previous iteration, vanilla_renderer.py:
// 111-117) and the roulette (vanilla_renderer.py:50-57).  false: the path ends here.
template <int BM, int SM, int TEX>
APT_D bool open_vertex(const ShadeArgs3* A_, Vertex& vx, Philox& rng, int prim, f3 rec_kd, uint32_t meta, float ray_pdf, float2 uv, int bounce) {
    const bool was_spec = (meta >> 24) & 1u;
    if (BM == 0x002) vx.bx.k_d = rec_kd;               // Lambertian-only scenes: type 1, not delta, not a BSDF (vertex_reset), colour from the record
    else vx.bx = ld_bxdf_lane((A_->sc).bxdf + vx.it.obj_id);
    if (TEX && (A_->sc).tex_i != nullptr) {                // the scene declares image textures (TEX kernels only)
        f3 tx;
        if (bounce == 0) {                           // PathTracer.process_ns, applied to the camera ray's hit only (vanilla_renderer.py:42)
            if (get_uv_item((A_->sc), 1, vx.it.obj_id, prim, uv.x, uv.y, tx)) { m33 R; rotation_between(mk3(0.f, 1.f, 0.f), vx.it.n_g, R); vx.it.n_s = mul(R, tx); }
            if (get_uv_item((A_->sc), 2, vx.it.obj_id, prim, uv.x, uv.y, tx)) vx.it.n_s = delocalize(vx.it.n_s, tx);
        }
        // it.tex (vanilla_renderer.py:66): every surface model reads its diffuse colour as select(tex invalid, k_d, tex)
        // and nothing else reads k_d on the device, so a valid lookup simply replaces this path's copy of k_d
        if (get_uv_item((A_->sc), 0, vx.it.obj_id, prim, uv.x, uv.y, tx)) vx.bx.k_d = tx;
    }
    const uint32_t lp = vx.id & ((1u << (A_->p).pix_bits) - 1u), s = vx.id >> (A_->p).pix_bits;
    vx.l_off = (s * (uint32_t)(A_->p).npix + lp) << 2;
    vx.draw0 = meta & 0xffffu;
    rng_init(rng, ((A_->p).world == 1) ? lp : ldq((A_->p).pix_key, lp << 2), (A_->p).seed, (uint32_t)((A_->p).cnt_base + (int)s + 1), vx.draw0);
    if (bounce > 0 && (A_->p).use_mis) {
        float e_pdf = 0.0f;
        if (vx.hit_light >= 0 && vx.bx.is_delta == 0 && !was_spec) e_pdf = emitter_solid_angle_pdf((A_->sc).src[vx.hit_light], vx.it, vx.d);
        vx.emission_weight = balance(ray_pdf, e_pdf);
    }
    if (!(SM & 2)) rng_open(rng);                   // no area lights: a shade with one light sample draws at most five numbers (rng.hpp)
    if ((A_->p).use_rr) {
        float mx = max3(vx.thr);
        if (mx < (A_->p).rr_threshold && bounce >= (A_->p).rr_bounce_th) {
            if (rng_float(rng) > mx) return false;
            vx.thr = vx.thr * srcp(mx + 1e-7f);
        }
    } else if (max3(vx.thr) < 1e-4f) return false;
    return true;
}
// Step 2, once per light sample (sample_light, path_tracer.py:537-554; vanilla_renderer.py:68-95): `want` - the sample is worth a shadow
// ray of direction `dir`, length `dist`, carrying `contrib`; `poisoned` - its MIS weight `mis_w` is NaN (the caller stores it: upstream
// the weight multiplies the sample even when the shadow ray is occluded, 0 * NaN, so it poisons the whole pixel-sample, which is zeroed
// at the end - reproduced without tracing).  LANE_SRC: whole-record emitter loads (the class kernels only: in C2's traced kernel their
// sixteen registers cost the fourth wave, 127 -> 132 VGPRs).
struct LightSample { bool want, sampled, poisoned; f3 dir, contrib; float dist, mis_w; };
template <int BM, int SM, bool LANE_SRC>
APT_D LightSample sample_light(const ShadeArgs3* A_, Vertex& vx, Philox& rng, const EmitterGeom& geom, const DevSrc src_only, bool active, bool& break_flag) {
    LightSample ls; ls.want = ls.sampled = ls.poisoned = false; ls.dir = ls.contrib = splat3(0.f); ls.dist = 0.f; ls.mis_w = 1.0f;
    if (!active || break_flag) return ls;
    const int ns = (A_->sc).n_sources;                    // wave-uniform: one light needs no modulo
    int sidx = rng_int(rng);                            // one int is always drawn
    sidx = (ns == 1) ? 0 : pymod(sidx, ns);
    float emitter_pdf = (A_->p).inv_ns;
    if (vx.hit_light >= 0) {
        if (ns <= 1) { break_flag = true; return ls; }
        sidx = rng_int(rng);
        sidx = (ns == 2) ? 0 : pymod(sidx, ns - 1);
        if (sidx >= vx.hit_light) sidx += 1;
        emitter_pdf = (A_->p).inv_ns1;
    }
    DevSrc src = src_only;
    if (ns != 1) src = LANE_SRC ? ld_src_lane((A_->sc).src + sidx) : (A_->sc).src[sidx];
    f3 shadow_int; float direct_pdf;
    const f3 to_emitter = emitter_sample_hit<SM>(src, geom, vx.hit_point, rng, shadow_int, direct_pdf) - vx.hit_point;
    ls.dist = fnorm(to_emitter);
    ls.dir = fdiv3(to_emitter, ls.dist);
    ls.sampled = true;
    const f3 direct_spec = surface_eval<BM>(vx.bx, vx.it, vx.d, ls.dir, (A_->sc).world_ior, (A_->p).two_sides);
    if ((A_->p).use_mis && !(src.bool_bits & 0x01)) ls.mis_w = balance(emitter_pdf * direct_pdf, surface_pdf<BM>(vx.bx, vx.it, ls.dir, vx.d, (A_->sc).world_ior, (A_->p).two_sides));
    if (isnan(ls.mis_w)) { ls.poisoned = true; return ls; }
    f3 c = (direct_spec * shadow_int) * ls.mis_w;
    if (ns != 1) c = fdiv3(c, emitter_pdf);               // one light: the pdf is exactly 1 and x / 1 == x (wave-uniform branch)
    ls.contrib = (c * (A_->p).inv_S) * vx.thr;
    ls.want = !(ls.contrib.x == 0.f && ls.contrib.y == 0.f && ls.contrib.z == 0.f);
    return ls;
}
// Step 3: emission of the surface the vertex is on (vanilla_renderer.py:99-104; handed to `gather`), then the continuation direction and
// the throughput behind it (vanilla_renderer.py:106-110).  The emission has to stay AFTER the light sampling: with two-sided BRDFs the
// evaluation there flips it.n_s in place, upstream as here, and eval_le sees the flipped normal.
// (pdf == 0 - a cosine-hemisphere draw of exactly 0, one in 2^24 - makes the throughput spec / 0: +inf when the rounding residue of n_s . out is
// positive, NaN when it is not; upstream lets +inf through to the pixel and zeroes NaN.  The residue hangs on the last bits of the
// un-normalised interpolated vertex normal, i.e. of the barycentrics, which the product build's intersectors return to 1e-6 and not
// to the bit: DESIGN.md section 5 "non-finite pixels".  Re-sampling such a vertex here with the reference's own triangle test was
// measured: it costs the Lambertian kernel its fourth wave per SIMD, 122 -> 130 / 158 VGPRs inline / as a loop.)
template <int BM, int SM, typename Gather>
APT_D f3 emit_and_scatter(const ShadeArgs3* A_, Vertex& vx, Philox& rng, float& new_pdf, bool& is_spec, Gather&& gather) {
    if ((SM & 2) && vx.hit_light >= 0) {
        const f3 emit_int = emitter_eval_le((A_->sc).src[vx.hit_light], vx.hit_point - vx.o, vx.it.n_s);
        if (!(emit_int.x == 0.f && emit_int.y == 0.f && emit_int.z == 0.f)) gather((emit_int * vx.emission_weight) * vx.thr);
    }
    f3 spec;
    const f3 new_d = surface_sample<BM>(vx.bx, vx.it, vx.d, (A_->sc).world_ior, (A_->p).two_sides, rng, spec, new_pdf, is_spec);
    vx.thr = vx.thr * fdiv3(spec, new_pdf);
    return new_d;
}
// a shadow-queue entry (k_shadow / shadow_flat_body read these planes)
APT_D void shadow_store(const Queues& q, uint32_t so, f3 o, f3 dir, float dist, f3 contrib) {
    st3q(q.sh_o, q.sh_cap, so, o); st3q(q.sh_d, q.sh_cap, so, dir); stq(q.sh_tmax, so, dist); st3q(q.sh_c, q.sh_cap, so, contrib);
}
struct ShadeTally { uint32_t shade, shadow, poison; };          // wave-uniform tallies (SGPRs); the RNG draws of a wave are tallied in LDS (s_draws): a per-lane tally would hold a VGPR for the whole kernel
APT_D void tally_flush(const ShadeTally& t, const uint32_t* s_draws, Counters* cnt, int sq) {
    flush_uniform(t.shade, &cnt->stats[sq][ST_SHADE]);
    flush_uniform(t.shadow, &cnt->stats[sq][ST_SHADOW]);
    if (lane_id() == 0 && s_draws[threadIdx.x >> 6]) atomicAdd(&cnt->stats[sq][ST_DRAWS], (unsigned long long)s_draws[threadIdx.x >> 6]);
    flush_uniform(t.poison, &cnt->stats[sq][ST_POISON]);
}
#if APT_FAST
// ---- the shade kernel that traces its own rays (Params::fused == 2: "rays traced in place" above).  Input and output are the packed
// records of Queues::tr; the path's radiance travels with it (Lc) and reaches its slot of L once, when the path ends.
// For a scene of a few dozen records the any-hit sweep of a shadow ray costs fewer issue slots than the round trip of its 44-byte queue
// entry through HBM plus the scattered read-modify-write of the path's radiance slot behind it: the vertex's light sample is swept right
// here (one ray per lane against two records per packed instruction, traverse.hpp flat_any1), after the continuation has been sampled and
// traced, when little else is live.  The rare rays whose answer needs the reference-order sweep (flat_needs_cull) still leave as
// shadow-queue entries, counted by n_fix_sh[cur], and are served by the next launch's prologue.
template <int BM, int SM, int TEX>
APT_D void shade_traced(args3_ptr A0, Counters* cnt, int cur, int bounce) {
    const ShadeArgs3* A_ = args_fresh(A0);                      // scene, parameters, queues: read through the kernel-argument segment, re-fetched per phase (APT_ARGS_PHASE)
    const int nxt = cur ^ 1;
    const SubLoop sl = sub_loop((A_->p).nq);
    const uint32_t qbase = (uint32_t)sl.q * (A_->p).subcap, sh_qbase = (uint32_t)sl.q * (A_->q).sh_subcap;
    uint32_t* next_counter = &cnt->n_tr[(bounce + 1) % 3][0][sl.q * CNT_PAD];      // (the first queue's tail)
    if (sl.first == 0 && threadIdx.x == 0) {                   // (read by the previous bounce, appended to by the next one)
        cnt->n_tr[(bounce + 2) % 3][0][sl.q * CNT_PAD] = 0;
        cnt->n_tr[(bounce + 2) % 3][(A_->q).tr_ncls][sl.q * CNT_PAD] = 0;      // (the staging queue's tail)
    }
    const EmitterGeom geom = {(A_->sc).precom, (A_->sc).normals, (A_->sc).obj_info};
    __shared__ uint32_t s_draws[BLOCK / 64];
    if (lane_id() == 0) s_draws[threadIdx.x >> 6] = 0;
    ShadeTally tl = {0u, 0u, 0u};
    uint32_t t_traced = 0, t_lit = 0, t_extend = 0;
#ifdef APT_NEAR_STATS
    uint32_t t_near = 0;
#endif
    // The next row's hit primitive is requested one row ahead (PFP: one register), so that a row's shading record can be requested together
    // with its queue record instead of a round trip after it.  (Rounds 2-4 prefetched the Lambertian kernel's whole record: thirteen
    // registers.  Without them the kernel allocates 91 VGPRs - five waves per SIMD instead of four - and three render lanes gain 6 %:
    // C2 4 310 -> 4 560 Msamples/s on the same box.)
    constexpr bool PFP = TEX == 0;
    static_assert(APT_FLAT_MAX_PRIMS < (int)TR_NO_PRIM, "the packed record keeps the hit primitive in 8 bits");
    const float4* trA = (A_->q).tr[cur][0]; const float4* trB = (A_->q).tr[cur][1]; const float4* trC = (A_->q).tr[cur][2]; const float4* trD = (A_->q).tr[cur][3];
    fix_prologue((A_->sc), (A_->p), (A_->q), cnt, cur, sl.q, bounce);       // before the queue's length is read: the prologue may append to it
    const uint32_t n = __hip_atomic_load(&cnt->n_tr[bounce % 3][0][sl.q * CNT_PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t pf_pm = 0;                                        // (the packed word: primitive, draw index, specular flag)
    auto prefetch_prim = [&](uint32_t b) { pf_pm = ldq(reinterpret_cast<const uint32_t*>(trB), ((qbase + min(b + threadIdx.x, n - 1u)) << 4) + 12u); };      // (lanes past the end re-read the last entry: never used)
    if (PFP && n > 0) prefetch_prim(sl.first);
    for (uint32_t base = sl.first; base < n; base += sl.stride) {
        APT_ARGS_PHASE();
        const uint32_t pos = base + threadIdx.x, idx = qbase + pos;                    // (unsorted renders: one queue for the scene)
        bool alive = pos < n;
        const uint32_t cu_pm = pf_pm;
        float4 cu_ra = make_float4(0.f, 0.f, 0.f, 0.f), cu_rb = cu_ra;
        if (PFP) { const int rp = max(tr_prim(cu_pm), 0); cu_ra = (A_->sc).prim_shade[2 * rp]; cu_rb = (A_->sc).prim_shade[2 * rp + 1]; prefetch_prim(base + sl.stride); }
        Vertex vx; vertex_reset(vx);
        Philox rng; rng_init(rng, 0u, 0u, 0u, 0u);
        f3 Lc = splat3(0.f);                                   // the radiance the path has gathered so far (camera rays carry none: nothing is read at bounce 0)
        const bool entry = alive;
        float ray_pdf = 1.f;
        if (alive && bounce > 0) { const float4 dd = ldq(trD, idx << 4); Lc = mk3(dd.x, dd.y, dd.z); if (SM & 2) ray_pdf = dd.w; }
        if (alive) {
            const float4 a = ldq(trA, idx << 4), b_ = ldq(trB, idx << 4), c = ldq(trC, idx << 4);
            vx.o = mk3(a.x, a.y, a.z); vx.d = mk3(b_.x, b_.y, b_.z); vx.thr = mk3(c.x, c.y, c.z); vx.id = __float_as_uint(c.w);
            const uint32_t pm = PFP ? cu_pm : __float_as_uint(b_.w);
            const int prim = tr_prim(pm);
            if (prim < 0) alive = false;                         // nothing hit: path ends (vanilla_renderer.py:49)
            else {
                f3 rec_kd; float2 uv; uv.x = uv.y = 0.f;
                if ((A_->sc).has_vn || (TEX && (A_->sc).tex_i != nullptr)) uv = ldq((A_->q).tr_uv[cur], idx << 3);      // otherwise nobody reads the barycentrics (and nobody wrote them)
                if (PFP) build_hit_rec((A_->sc), cu_ra, cu_rb, prim, a.w, uv.x, uv.y, vx.o, vx.d, vx.it, vx.hit_light, rec_kd);
                else build_hit((A_->sc), prim, a.w, uv.x, uv.y, vx.o, vx.d, vx.it, vx.hit_light, rec_kd);
                alive = open_vertex<BM, SM, TEX>(A_, vx, rng, prim, rec_kd, tr_meta(pm, (uint32_t)bounce), ray_pdf, uv, bounce);
            }
        }
        tl.shade += wave_count(alive);
#ifdef APT_NEAR_STATS      // diagnostic build (tools/gpu_near_probe.py): shaded vertices that sit within 2e-3 of the vertex before them - rays that re-hit the surface they left
        t_near += wave_count(alive && bounce > 0 && vx.it.min_depth < 2e-3f);
#endif
        if (alive) vx.hit_point = vx.d * vx.it.min_depth + vx.o;

        APT_ARGS_PHASE();
        // ---- next-event estimation: the vertex's light sample waits in registers and is swept at the end of the row, when little else is live
        bool break_flag = false;
        DevSrc src_only;                                      // the scene's only light, read once through the scalar path
        if ((A_->sc).n_sources == 1) src_only = ld_src_uniform((A_->sc).src);
        const LightSample f = sample_light<BM, SM, false>(A_, vx, rng, geom, src_only, alive, break_flag);      // (one light sample per vertex: api.hip, Params::fused)
        if (f.poisoned) Lc = splat3(f.mis_w);
        tl.shadow += wave_count(f.sampled); tl.poison += wave_count(f.poisoned);
        APT_ARGS_PHASE();
        // ---- emission of the surface we are on, then the continuation
        bool cont = false, is_spec = false;
        f3 new_d = mk3(0.f, 1.f, 0.f);
        float new_pdf = 1.f;
        if (alive) {
            new_d = emit_and_scatter<BM, SM>(A_, vx, rng, new_pdf, is_spec, [&](f3 add) { Lc = Lc + add; });
            cont = (bounce + 1) < (A_->p).max_bounce;
        }
        if (rng.draw != vx.draw0) atomicAdd(&s_draws[threadIdx.x >> 6], rng.draw - vx.draw0);      // also paths that died in the roulette
        APT_ARGS_PHASE();
        // the continuation ray meets the scene's records here.  Only rays that hit something (or whose answer is left to the reference-order
        // code: listed, with a provisional record) enter the next queue; the tail atomic is on its way while the light sample is swept below.
        float tr_t = 0.f; int tr_hit = -1; float tr_u = 0.f, tr_v = 0.f; int tr_q = -1;      // tr_q: the queue the record joins (-1: none)
        TrAppend tr_app; tr_app.raw = 0u; tr_app.rank = 0u;
        {
            int tr_idx = -1, tr_run = -1, hit_cls = 0;
            if (__any(cont)) tr_idx = flat_closest1((A_->sc).flat, vx.hit_point, new_d, 1e7f, tr_t, tr_run);
            const bool tr_defer = cont && (tr_run >= 0 || flat_needs_cull((A_->sc).flat, new_d));
            t_extend += wave_count(cont);
            cont = cont && (tr_idx >= 0 || tr_defer);
            if (cont && !tr_defer) { HitRec hr; flat_resolve((A_->sc).flat, tr_idx, tr_t, vx.hit_point, new_d, hr, hit_cls); tr_hit = hr.prim; tr_u = hr.u; tr_v = hr.v; }
            tr_q = !cont ? -1 : (tr_defer ? 1 : 0);            // (the queue, or the staging area at the top of its sub-queue's region)
            { const Append a_ = append_issue(tr_q == 0, next_counter); tr_app.raw = a_.raw; tr_app.rank = rank_in(a_.m); }      // (a staged ray - rare - moves the staging queue's tail by itself, below)
        }
        APT_ARGS_PHASE();
        {
            // the row's light sample, swept in place; a ray that needs the reference-order sweep leaves as a shadow-queue entry for the next launch's prologue
            const bool defer = f.want && flat_needs_cull((A_->sc).flat, f.dir);
            bool occ = false;
            if (__any(f.want && !defer)) occ = flat_any1((A_->sc).flat, vx.hit_point, f.dir, (f.dist > 0.0f) ? f.dist - 1e-4f : 1e7f);
            if (__any(defer)) {
                const uint32_t spos = wave_append(defer, &cnt->n_fix_sh[cur][sl.q * CNT_PAD]);
                if (defer && spos < (A_->q).sh_subcap) { const uint32_t so = (sh_qbase + spos) << 2; shadow_store((A_->q), so, vx.hit_point, f.dir, f.dist, f.contrib); stq((A_->q).sh_id, so, vx.l_off); }
            }
            const bool traced = f.want && !defer;
            if (traced) {
                // (an occluded sample still enters upstream's sum as 0 * contribution: NaN for a non-finite one, see k_shadow)
                const bool weird = !(isfinite(f.contrib.x) && isfinite(f.contrib.y) && isfinite(f.contrib.z));
                if (!occ || weird) Lc = Lc + (occ ? f.contrib * 0.f : f.contrib);
            }
            t_traced += wave_count(f.want); t_lit += wave_count(traced && !occ);
        }
        uint32_t npos = (uint32_t)__builtin_amdgcn_readlane((int)tr_app.raw, 0) + tr_app.rank;
        if (__any(tr_q == 1)) { const uint32_t dpos = wave_append(tr_q == 1, next_counter + APT_MAX_NQ * CNT_PAD); if (tr_q == 1) npos = dpos; }
        if (cont) {
            const uint32_t slot = (tr_q == 1) ? qbase + (A_->p).subcap - 1u - npos : qbase + npos, so = slot << 4;      // (staged rays grow down from the top of the sub-queue's region)
            stq((A_->q).tr[nxt][0], so, make_float4(vx.hit_point.x, vx.hit_point.y, vx.hit_point.z, tr_t));
            stq((A_->q).tr[nxt][1], so, make_float4(new_d.x, new_d.y, new_d.z, __uint_as_float(tr_pack(tr_hit, rng.draw, is_spec))));
            stq((A_->q).tr[nxt][2], so, make_float4(vx.thr.x, vx.thr.y, vx.thr.z, __uint_as_float(vx.id)));
            stq((A_->q).tr[nxt][3], so, make_float4(Lc.x, Lc.y, Lc.z, new_pdf));
            if ((A_->sc).has_vn || (A_->sc).tex_i != nullptr) { float2 uv_; uv_.x = tr_u; uv_.y = tr_v; stq((A_->q).tr_uv[nxt], slot << 3, uv_); }
        }
        if (!cont && entry && !(Lc.x == 0.f && Lc.y == 0.f && Lc.z == 0.f)) {
            // the path ends here (nothing hit, roulette, last bounce): its radiance goes to its slot - added, not stored: the prologue may have put a deferred sample's share there already
            const uint32_t lp_ = vx.id & ((1u << (A_->p).pix_bits) - 1u), s_ = vx.id >> (A_->p).pix_bits;
            add_radiance((A_->q).L, (A_->p).cap, (s_ * (uint32_t)(A_->p).npix + lp_) << 2, Lc, true);
        }
    }
    flush_uniform(t_traced, &cnt->stats[sl.q][ST_SHADOW_TRACED]); flush_uniform(t_lit, &cnt->stats[sl.q][ST_LIT]);
    flush_uniform(t_extend, &cnt->stats[sl.q][ST_EXTEND]);
    tally_flush(tl, s_draws, cnt, sl.q);
#ifdef APT_NEAR_STATS
    flush_uniform(t_near, &cnt->stats[sl.q][14]);
#endif
}
#endif
// ---- the staged shade kernel: reads the extend stage's output - the SoA queues (unsorted renders: ShadeIn) or one packed class queue
// (CQ: Queues::cq, class in.cls) - and writes the next bounce's SoA queue and one shadow-queue entry per useful light sample.
// BM / SM: material and emitter masks of the scene (shading.hpp); code for absent models is compiled out.
// TEX: image-texture lookups.  Only the all-models kernel is instantiated with TEX = 1 (textured scenes run unsorted through
// it): inlined into the specialised kernels the lookup costs e.g. the mod-Phong class kernel its fourth wave per SIMD
// (126 -> 129 VGPRs) in every scene WITHOUT textures, and out of line it costs a call frame in scratch.
template <int BM, int SM, int TEX, bool CQ>
APT_D void shade_staged(args3_ptr A0, Counters* cnt, const ShadeIn& in, int cur, int bounce) {
    const ShadeArgs3* A_ = args_fresh(A0);                      // scene, parameters, queues: read through the kernel-argument segment, re-fetched per phase (APT_ARGS_PHASE)
    const int nxt = cur ^ 1;
    const SubLoop sl = sub_loop((A_->p).nq);
    const uint32_t n = in.counts[sl.q * CNT_PAD];
    const uint32_t qbase = (uint32_t)sl.q * (A_->p).subcap, sh_qbase = (uint32_t)sl.q * (A_->q).sh_subcap;
    uint32_t* next_counter = &cnt->n_active[nxt][sl.q * CNT_PAD];
    uint32_t* shadow_counter = &cnt->n_shadow[sl.q * CNT_PAD];
    const EmitterGeom geom = {(A_->sc).precom, (A_->sc).normals, (A_->sc).obj_info};
    __shared__ uint32_t s_draws[BLOCK / 64];
    if (lane_id() == 0) s_draws[threadIdx.x >> 6] = 0;
    ShadeTally tl = {0u, 0u, 0u};
#ifdef APT_NEAR_STATS
    uint32_t t_near = 0;
#endif
    constexpr bool PFP = TEX == 0;                            // the next row's hit primitive is requested one row ahead, as in shade_traced
    const uint32_t in_base = CQ ? (uint32_t)in.cls * (A_->p).cap + qbase : qbase;      // first slot of the queue this workgroup reads
    const float4* cqA = CQ ? (A_->q).cq[0] : nullptr; const float4* cqB = CQ ? (A_->q).cq[1] : nullptr; const float4* cqC = CQ ? (A_->q).cq[2] : nullptr; const float4* cqD = CQ ? (A_->q).cq[3] : nullptr;
    int pf_prim = -1;
    auto prefetch_prim = [&](uint32_t b) {
        const uint32_t ps = in_base + min(b + threadIdx.x, n - 1u);
        pf_prim = CQ ? ldq(reinterpret_cast<const int*>(cqB), (ps << 4) + 12u) : ldq(in.prim, ps << 2);
    };
    if (PFP && n > 0) prefetch_prim(sl.first);
    for (uint32_t base = sl.first; base < n; base += sl.stride) {
        APT_ARGS_PHASE();
        const uint32_t pos = base + threadIdx.x, idx = in_base + pos;
        bool alive = pos < n;
        const int cu_prim = pf_prim;
        float4 cu_ra = make_float4(0.f, 0.f, 0.f, 0.f), cu_rb = cu_ra;
        if (PFP) { const int rp = max(cu_prim, 0); cu_ra = (A_->sc).prim_shade[2 * rp]; cu_rb = (A_->sc).prim_shade[2 * rp + 1]; prefetch_prim(base + sl.stride); }
        Vertex vx; vertex_reset(vx);
        Philox rng; rng_init(rng, 0u, 0u, 0u, 0u);
        if (alive) {
            const uint32_t io = idx << 2;
            const int prim = PFP ? cu_prim : (CQ ? ldq(reinterpret_cast<const int*>(cqB), (idx << 4) + 12u) : ldq(in.prim, io));
            if (prim < 0) alive = false;                         // nothing hit: path ends (vanilla_renderer.py:49)
            else {
                uint32_t meta; float t_in, ray_pdf = 1.f; float2 uv; uv.x = uv.y = 0.f;
                if (CQ) {
                    const float4 a = ldq(cqA, idx << 4), b_ = ldq(cqB, idx << 4), c = ldq(cqC, idx << 4), dd = ldq(cqD, idx << 4);
                    vx.o = mk3(a.x, a.y, a.z); t_in = a.w; vx.d = mk3(b_.x, b_.y, b_.z); vx.thr = mk3(c.x, c.y, c.z); vx.id = __float_as_uint(c.w);
                    meta = __float_as_uint(dd.x); if (SM & 2) ray_pdf = dd.y; uv.x = dd.z; uv.y = dd.w;
                } else {
                    vx.o = ld3q(in.ray_o, (A_->p).cap, io); vx.d = ld3q(in.ray_d, (A_->p).cap, io); vx.thr = ld3q(in.thr, (A_->p).cap, io);
                    vx.id = ldq(in.id, io); meta = ldq(in.meta, io); t_in = ldq(in.t, io);
                    if (SM & 2) ray_pdf = ldq(in.pdf, io);          // (its only reader is the emission MIS weight: scenes without area lights never look at it)
                    if ((A_->sc).has_vn || (TEX && (A_->sc).tex_i != nullptr)) { uv.x = ldq(in.u, io); uv.y = ldq(in.v, io); }      // otherwise nobody reads the barycentrics (and the flat extend kernel does not write them)
                }
                f3 rec_kd;
                if (PFP) build_hit_rec((A_->sc), cu_ra, cu_rb, prim, t_in, uv.x, uv.y, vx.o, vx.d, vx.it, vx.hit_light, rec_kd);
                else build_hit((A_->sc), prim, t_in, uv.x, uv.y, vx.o, vx.d, vx.it, vx.hit_light, rec_kd);
                alive = open_vertex<BM, SM, TEX>(A_, vx, rng, prim, rec_kd, meta, ray_pdf, uv, bounce);
            }
        }
        tl.shade += wave_count(alive);
#ifdef APT_NEAR_STATS      // diagnostic build (tools/gpu_near_probe.py): shaded vertices that sit within 2e-3 of the vertex before them - rays that re-hit the surface they left
        t_near += wave_count(alive && bounce > 0 && vx.it.min_depth < 2e-3f);
#endif
        if (alive) vx.hit_point = vx.d * vx.it.min_depth + vx.o;

        APT_ARGS_PHASE();
        // ---- next-event estimation: one shadow-queue entry per useful light sample
        bool break_flag = false;
        DevSrc src_only;                                      // the scene's only light, read once through the scalar path
        if ((A_->sc).n_sources == 1) src_only = ld_src_uniform((A_->sc).src);
        // light samples by vertex: ONE queue-tail atomic per tile row for all S samples of every vertex
        uint32_t vbase = 0;
        if ((A_->p).nee_vm) vbase = wave_append(alive, shadow_counter);
        for (int s = 0; s < (A_->p).S; s++) {
            const LightSample ls = sample_light<BM, SM, CQ>(A_, vx, rng, geom, src_only, alive, break_flag);
            if (ls.poisoned) stL((A_->q).L, (A_->p).cap, vx.l_off, splat3(ls.mis_w));
            tl.shadow += wave_count(ls.sampled); tl.poison += wave_count(ls.poisoned);
            if ((A_->p).nee_vm) {
                const uint32_t so = (sh_qbase + (uint32_t)s * (A_->p).subcap + vbase) << 2;        // plane s of the sub-queue's region: consecutive lanes, consecutive entries
                if (ls.want) { st3q((A_->q).sh_d, (A_->q).sh_cap, so, ls.dir); stq((A_->q).sh_tmax, so, ls.dist); st3q((A_->q).sh_c, (A_->q).sh_cap, so, ls.contrib); }
                else if (alive) stq((A_->q).sh_tmax, so, -1.0f);                   // the vertex has no sample s worth tracing
                if (alive && s == 0) { stq((A_->q).sh_id, so, vx.l_off); st3q((A_->q).sh_o, (A_->q).sh_cap, so, vx.hit_point); }      // one radiance slot and ONE origin per vertex, kept with its first entry (the S samples leave from the same point: 12 bytes written and read once instead of S times)
            } else {
                const uint32_t spos = wave_append(ls.want, shadow_counter);
                if (ls.want && spos < (A_->q).sh_subcap) {
                    const uint32_t so = (sh_qbase + spos) << 2;
                    shadow_store((A_->q), so, vx.hit_point, ls.dir, ls.dist, ls.contrib);
                    stq((A_->q).sh_id, so, vx.l_off | (((A_->p).l_planes > 1) ? (uint32_t)s : 0u));
                }
            }
        }
        APT_ARGS_PHASE();
        // ---- emission of the surface we are on, then the continuation
        bool cont = false, is_spec = false;
        f3 new_d = mk3(0.f, 1.f, 0.f);
        float new_pdf = 1.f;
        if (alive) {
            new_d = emit_and_scatter<BM, SM>(A_, vx, rng, new_pdf, is_spec, [&](f3 add) { add_radiance((A_->q).L, (A_->p).cap, vx.l_off, add, true); });      // (nothing else touches the path's slot while its shade kernel runs)
            cont = (bounce + 1) < (A_->p).max_bounce;
        }
        if (rng.draw != vx.draw0) atomicAdd(&s_draws[threadIdx.x >> 6], rng.draw - vx.draw0);      // also paths that died in the roulette
        APT_ARGS_PHASE();
        const uint32_t npos = wave_append(cont, next_counter);
        if (cont) {
            const uint32_t so = (qbase + npos) << 2;
            st3q((A_->q).ray_o[nxt], (A_->p).cap, so, vx.hit_point);
            st3q((A_->q).ray_d[nxt], (A_->p).cap, so, new_d);
            st3q((A_->q).thr[nxt], (A_->p).cap, so, vx.thr);
            stq((A_->q).id[nxt], so, vx.id);
            stq((A_->q).meta[nxt], so, pack_meta(rng.draw, (uint32_t)(bounce + 1), is_spec));
            if (SM & 2) stq((A_->q).pdf[nxt], so, new_pdf);
        }
    }
    tally_flush(tl, s_draws, cnt, sl.q);
#ifdef APT_NEAR_STATS
    flush_uniform(t_near, &cnt->stats[sl.q][14]);
#endif
}
template <int BM, int SM, int TEX = 0>
__global__ void __launch_bounds__(BLOCK) k_shade(DevScene sc, Params p, Queues q, Counters* cnt, ShadeIn in, int cur, int bounce) {
    shade_staged<BM, SM, TEX, false>(kernel_args3(), cnt, in, cur, bounce);
}
#if APT_FAST
template <int BM, int SM, int TEX = 0>
__global__ void __launch_bounds__(BLOCK) k_shade_traced(DevScene sc, Params p, Queues q, Counters* cnt, int cur, int bounce) {
    shade_traced<BM, SM, TEX>(kernel_args3(), cnt, cur, bounce);
}
// The Lambertian / point-light specialisation (C1 / C2) under a register bound: with round 6's float transcendentals it allocates 77 VGPRs
// (six waves per SIMD); asked for seven the allocator finds 72 without a spill (eight: 64 and 24 bytes of scratch).
#ifndef APT_TRACED_WAVES
#define APT_TRACED_WAVES 7
#endif
template <int BM, int SM>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(APT_TRACED_WAVES, APT_TRACED_WAVES))) k_shade_traced_lean(DevScene sc, Params p, Queues q, Counters* cnt, int cur, int bounce) {
    shade_traced<BM, SM, 0>(kernel_args3(), cnt, cur, bounce);
}
#endif

// ---- class kernels in groups: ONE launch shades several material classes.
// A class-sorted bounce was one launch per class (C5: 8 x 16 per batch), and a launch costs its lane 15-30 us however short its queue -
// the pipeline drains, the next grid is dispatched, 1024 workgroups read their queue lengths: most of C5's shade time, a tenth of C3's.
// The classes of a bounce are independent, so a group kernel walks the class queues of its members one after the other - every workgroup
// its sub-queue of class A, then of class B, ... with no barrier in between: a workgroup that runs out of A entries starts on B while
// others still shade A - and the launch boundary between them is gone.  A kernel's register allocation is the maximum over its members',
// so the groups follow the footprints (api.hip kClassGroup).  Round 5 had three - up to 128 VGPRs / four waves per SIMD, up to 168 / three,
// beyond / two (modified Phong and Fresnel blend with their double-precision pow: 201-217).  With the product build's float transcendentals
// and 1-ulp divisions (round 6) no class kernel allocates more than 125, so there are TWO: the lean classes - Lambertian, delta, lobe-free
// Blinn-Phong, Lambertian transmission: 93-96 VGPRs, held to FIVE waves per SIMD - and everything else (Blinn-Phong, Oren-Nayar, thin coat,
// microfacet, modified Phong, Fresnel blend: 111-125, four waves).
// Members absent from the scene are skipped by a wave-uniform test (GroupIn::cls < 0).  Per vertex nothing changes: same class code, same
// queues, same order inside a queue - images and statistics are those of the one-launch-per-class schedule bit for bit (GPU test).
#define APT_GROUP_SLOTS 6
struct GroupIn { const uint32_t* counts[APT_GROUP_SLOTS]; int cls[APT_GROUP_SLOTS]; };     // per member: the class queue's per-sub-queue entry counts, its compact class id (-1: not in this scene)
template <int SM, int WAVES, int B0, int B1, int B2, int B3, int B4 = 0, int B5 = 0>
__global__ void __launch_bounds__(BLOCK, WAVES) k_shade_group(DevScene sc, Params p, Queues q, Counters* cnt, GroupIn g, int cur, int bounce) {
    ShadeIn in = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if constexpr (B0 != 0) if (g.cls[0] >= 0) { in.counts = g.counts[0]; in.cls = g.cls[0]; shade_staged<B0, SM, 0, true>(kernel_args3(), cnt, in, cur, bounce); }
    if constexpr (B1 != 0) if (g.cls[1] >= 0) { in.counts = g.counts[1]; in.cls = g.cls[1]; shade_staged<B1, SM, 0, true>(kernel_args3(), cnt, in, cur, bounce); }
    if constexpr (B2 != 0) if (g.cls[2] >= 0) { in.counts = g.counts[2]; in.cls = g.cls[2]; shade_staged<B2, SM, 0, true>(kernel_args3(), cnt, in, cur, bounce); }
    if constexpr (B4 != 0) if (g.cls[4] >= 0) { in.counts = g.counts[4]; in.cls = g.cls[4]; shade_staged<B4, SM, 0, true>(kernel_args3(), cnt, in, cur, bounce); }
    if constexpr (B5 != 0) if (g.cls[5] >= 0) { in.counts = g.counts[5]; in.cls = g.cls[5]; shade_staged<B5, SM, 0, true>(kernel_args3(), cnt, in, cur, bounce); }
    if constexpr (B3 != 0) if (g.cls[3] >= 0) { in.counts = g.counts[3]; in.cls = g.cls[3]; shade_staged<B3, SM, 0, true>(kernel_args3(), cnt, in, cur, bounce); }
}
