// rng.hpp — counter-based RNG for the path tracer: Philox-4x32-10 (Salmon et al., SC'11).
//
// Replaces the reference's stateful `ti.random` (third-party Taichi; call sites listed in
// SURVEY.md A.4).  Stream definition (the CPU checker and the golden generator implement the same stream):
//   key     = (global pixel index x*H + y, seed)
//   counter = (sample counter `cnt` of that pixel-sample, draw_index / 4, 0, 0)
//   draw d  = word (d & 3) of that block; floats use the top 24 bits -> [0,1); ints are the raw word
// A path carries only its draw index; the last generated block is cached in registers.
#pragma once
#include "vec.hpp"
#include <stdint.h>

struct Philox {
    uint32_t key0, key1, ctr0;
    uint32_t draw;
    uint32_t blk;        // block index held in c[] (0xffffffff = none)
    uint32_t nblk;       // block index held in n[] (0xffffffff = none): only ever set by rng_open
    uint32_t c[4], n[4];
};

APT_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
    // A stage calls this from several inlined draw sites with the same key.  Left alone, the compiler shares the ten
    // per-round keys (k0 + r * 0x9E3779B9) between the sites and keeps them in ~10 VGPRs for the whole kernel; making the
    // key opaque per call re-derives them with ten adds each time and frees those registers.
    asm volatile("" : "+v"(k0));
#endif
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

APT_HD void rng_init(Philox& r, uint32_t pixel, uint32_t seed, uint32_t sample, uint32_t draw) {
    r.key0 = pixel; r.key1 = seed; r.ctr0 = sample; r.draw = draw; r.blk = 0xffffffffu; r.nblk = 0xffffffffu;
    r.c[0] = r.c[1] = r.c[2] = r.c[3] = 0u;
    r.n[0] = r.n[1] = r.n[2] = r.n[3] = 0u;
}
APT_HD uint32_t rng_u32(Philox& r) {
    const uint32_t d = r.draw++;
    const uint32_t b = d >> 2;
    const bool step = b != r.blk;
    if (step && b != r.nblk) { philox4x32_10(r.ctr0, b, 0u, 0u, r.key0, r.key1, r.n); r.nblk = b; }      // not opened ahead: generate here
    if (step) { r.c[0] = r.n[0]; r.c[1] = r.n[1]; r.c[2] = r.n[2]; r.c[3] = r.n[3]; r.blk = b; r.nblk = 0xffffffffu; }
    const uint32_t w = d & 3u, c0 = r.c[0], c1 = r.c[1], c2 = r.c[2], c3 = r.c[3];     // select without dynamic register indexing (of VALUES: a conditional over the array's lvalues is a pointer phi, which can keep the block in scratch)
    return (w == 0u) ? c0 : ((w == 1u) ? c1 : ((w == 2u) ? c2 : c3));
}
// For a stage that draws at most five numbers per path: both blocks those draws can touch, generated now, unconditionally.  A generation
// computes every lane's own block in one pass, but left to the draw sites it runs at every site where ANY lane steps into a new block,
// and the lanes of an unsorted queue sit at unrelated offsets of their streams: 3.4 passes per shade on the Cornell box where two
// serve every lane (k_shade, point lights, one light sample: 3.03 -> 2.83 ms per 64 spp).  Stages that draw more keep the lazy
// per-site generation: there the second buffer only costs registers (measured: C3 13.2 -> 14.7 ms with it).
APT_HD void rng_open(Philox& r) {
    const uint32_t b = r.draw >> 2;
    philox4x32_10(r.ctr0, b, 0u, 0u, r.key0, r.key1, r.c); r.blk = b;
    philox4x32_10(r.ctr0, b + 1u, 0u, 0u, r.key0, r.key1, r.n); r.nblk = b + 1u;
}
APT_HD float rng_float(Philox& r) { return (float)(rng_u32(r) >> 8) * (1.0f / 16777216.0f); }
APT_HD int32_t rng_int(Philox& r) { return (int32_t)rng_u32(r); }
// Python-style modulo: the reference's `ti.random(int) % n` is non-negative
APT_HD int pymod(int a, int n) { int m = a % n; return (m < 0) ? m + n : m; }
