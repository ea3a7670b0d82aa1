// unit_kernels.hpp - the entry kernels behind the C-ABI's unit functions (apt_occluded, apt_rng_stream, apt_bxdf_*, apt_texture_query,
// apt_emitter_probe, apt_clock_probe): single device functions run on explicit inputs, for the parity tests and the bench's clock check.
#pragma once
#include "shade_stage.hpp"

// Shader clock under load: every wave of a full grid runs a dependent FMA chain for a fixed number of iterations and reports the
// cycle counter (s_memtime-class counter, shader clock) against the constant 100 MHz wall clock.  out[2*w] = cycles, out[2*w+1] = ticks.
__global__ void __launch_bounds__(BLOCK) k_clock_probe(int iters, float seed, unsigned long long* out, float* sink) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float a = seed + (float)threadIdx.x, b = 1.000001f;
    for (int i = 0; i < iters; i++) { a = __builtin_fmaf(a, b, 0.5f); b = __builtin_fmaf(b, 0.999999f, 1e-7f); }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (a == 12345.678f) sink[0] = a + b;                  // keeps the chain alive
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
        out[2 * w] = c1 - c0; out[2 * w + 1] = w1 - w0;
    }
}

// ------------------------------------------------------- unit entry kernels
template <int MODE>
__global__ void __launch_bounds__(TRACE_NT(MODE)) k_occluded(DevScene sc, uint32_t n, const float* o_, const float* d_, const float* tmax, int* occ, LdsPlan plan) {
    for (uint32_t base = blockIdx.x * TRACE_NT(MODE); base < n; base += gridDim.x * TRACE_NT(MODE)) {
        const uint32_t pos = base + threadIdx.x;
        const bool valid = pos < n;
        const uint32_t idx = valid ? pos : n - 1;
        f3 o = mk3(o_[idx], o_[n + idx], o_[2 * n + idx]), d = mk3(d_[idx], d_[n + idx], d_[2 * n + idx]);
        HitRec rec; rec.t = (tmax[idx] > 0.0f) ? tmax[idx] - 1e-4f : 1e7f; rec.prim = -1; rec.u = rec.v = 0.f;
        const bool hit = (MODE == 0) ? traverse<true>(sc.bvh, make_stack(plan), o, d, rec)
                       : (MODE == 1) ? sweep_any(sc.sweep, o, d, rec)
                                     : sweep_tile<true, APT_TILE_NT>(sc.sweep, o, d, rec, valid, reinterpret_cast<float*>(s_dyn));
        if (valid) occ[idx] = hit ? 1 : 0;
    }
}
__global__ void k_rng_stream(uint32_t pixel, uint32_t seed, uint32_t sample, int n, uint32_t* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        Philox r; rng_init(r, pixel, seed, sample, 0u);
        for (int k = 0; k < n; k++) out[k] = rng_u32(r);
    }
}
// BxDF eval / pdf / sample on explicit inputs, RNG = Philox stream keyed by (test index, seed, 1)
// in : per test 10 floats n_s n_g incid [+ out for eval]; bx: one DevBxdf per test
__global__ void k_bxdf_eval(int n, const DevBxdf* bx, const float* in, float world_ior, float* out4) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* x = in + 12 * k;
    Hit it; it.obj_id = 0; it.prim_id = 0; it.min_depth = 1.f; it.n_s = ld3(x); it.n_g = ld3(x + 3);
    f3 wi = ld3(x + 6), wo = ld3(x + 9);
    DevBxdf b = bx[k];
    f3 e = surface_eval<APT_BX_ALL>(b, it, wi, wo, world_ior, 0);
    float pdf = surface_pdf<APT_BX_ALL>(b, it, wo, wi, world_ior, 0);
    out4[4 * k] = e.x; out4[4 * k + 1] = e.y; out4[4 * k + 2] = e.z; out4[4 * k + 3] = pdf;
}
__global__ void k_bxdf_sample(int n, const DevBxdf* bx, const float* in, float world_ior, uint32_t seed, float* out9) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* x = in + 12 * k;
    Hit it; it.obj_id = 0; it.prim_id = 0; it.min_depth = 1.f; it.n_s = ld3(x); it.n_g = ld3(x + 3);
    f3 wi = ld3(x + 6);
    DevBxdf b = bx[k];
    Philox r; rng_init(r, (uint32_t)k, seed, 1u, 0u);
    f3 spec; float pdf; bool sp;
    f3 dir = surface_sample<APT_BX_ALL>(b, it, wi, world_ior, 0, r, spec, pdf, sp);
    float* o = out9 + 9 * k;
    o[0] = dir.x; o[1] = dir.y; o[2] = dir.z; o[3] = spec.x; o[4] = spec.y; o[5] = spec.z; o[6] = pdf; o[7] = sp ? 1.f : 0.f; o[8] = (float)r.draw;
}
__global__ void k_texture_probe(DevScene sc, int n, const int* map_obj, const float* uv, float* out3) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    f3 r = texture_query(sc, map_obj[2 * k], map_obj[2 * k + 1], uv[2 * k], uv[2 * k + 1]);
    out3[3 * k] = r.x; out3[3 * k + 1] = r.y; out3[3 * k + 2] = r.z;
}
// emitter sample_hit / eval_le / solid_angle_pdf on explicit inputs: in = src index, hit_pos, normal, ray_d, min_depth (11 floats)
__global__ void k_emitter_probe(DevScene sc, int n, const float* in, uint32_t seed, float* out12) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* x = in + 11 * k;
    const DevSrc s = sc.src[(int)x[0]];
    const EmitterGeom geom = {sc.precom, sc.normals, sc.obj_info};
    Philox r; rng_init(r, (uint32_t)k, seed, 1u, 0u);
    f3 inten; float pdf;
    f3 pos = emitter_sample_hit<APT_SRC_ALL>(s, geom, ld3(x + 1), r, inten, pdf);
    Hit it; it.obj_id = 0; it.prim_id = 0; it.n_s = it.n_g = ld3(x + 4); it.min_depth = x[10];
    f3 rd = ld3(x + 7);
    f3 le = emitter_eval_le(s, rd * x[10], ld3(x + 4));
    float sap = emitter_solid_angle_pdf(s, it, rd);
    float* o = out12 + 12 * k;
    o[0] = pos.x; o[1] = pos.y; o[2] = pos.z; o[3] = inten.x; o[4] = inten.y; o[5] = inten.z; o[6] = pdf; o[7] = (float)r.draw;
    o[8] = le.x; o[9] = le.y; o[10] = le.z; o[11] = sap;
}
