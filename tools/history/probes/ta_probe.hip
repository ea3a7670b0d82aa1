// ta_probe.hip - how the vector-memory pipe of gfx950 prices divergent 16-byte loads (DESIGN.md "what the walk is bound by").
// Each lane makes ITER dependent rounds of NLOAD dwordx4 loads from a 6 MB table (L2 / MALL resident), with one of these address patterns:
//   0: per lane 5 consecutive 16-byte pieces of a random 80-byte record (the BVH node fetch as the walk does it: 5 instructions, 64 records)
//   1: transposed: the wave's 64 records' 320 pieces dealt to lanes in order (lane L, instruction j -> piece (L + 64 j) % 5 of record (L + 64 j) / 5)
//   2: like 1 but records 128-byte aligned (stride 128, 80 used)
//   3: one 16-byte piece per lane from a random record (64 distinct lines per instruction), 5 instructions -> the "per address" price
//   4: fully coalesced (lane L reads piece L of a random 1 KB block)
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/ta_probe.hip -o /tmp/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void __launch_bounds__(256) probe(const uint4* tab, uint32_t n_rec, int iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = wave * 0x9e3779b9u + 1u;
    const char* base = (const char*)tab;
    for (int it = 0; it < iters; it++) {
        uint32_t sum = 0;
        if (MODE == 0) {
            const uint32_t rec = hash(acc + lane * 0x85ebca6bu) % n_rec;
#pragma unroll
            for (int j = 0; j < 5; j++) { const uint4 v = *(const uint4*)(base + (size_t)rec * 80u + 16u * j); sum += v.x ^ v.w; }
        } else if (MODE == 1 || MODE == 2) {
            const uint32_t stride = MODE == 1 ? 80u : 128u;
            const uint32_t nr = MODE == 1 ? n_rec : (n_rec * 80u) / 128u;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint32_t g = lane + 64u * j, r = g / 5u, c = g - 5u * r;
                const uint32_t rec = hash(acc + r * 0x85ebca6bu) % nr;
                const uint4 v = *(const uint4*)(base + (size_t)rec * stride + 16u * c); sum += v.x ^ v.w;
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 5; j++) { const uint32_t rec = hash(acc + lane * 0x85ebca6bu + j * 0xc2b2ae35u) % n_rec; const uint4 v = *(const uint4*)(base + (size_t)rec * 80u); sum += v.x ^ v.w; }
        } else {
#pragma unroll
            for (int j = 0; j < 5; j++) { const uint32_t blk = hash(acc + j * 0xc2b2ae35u) % (n_rec * 80u / 1024u); const uint4 v = *(const uint4*)(base + (size_t)blk * 1024u + 16u * lane); sum += v.x ^ v.w; }
        }
        // wave-uniform dependency for the next round (as a walk's next node depends on this one)
        acc = hash(acc ^ (sum & 1u) ^ (uint32_t)it);
        acc = __builtin_amdgcn_readfirstlane(acc);
    }
    if (acc == 0x12345u) out[0] = acc;
}
int main(int argc, char** argv) {
    const uint32_t n_rec = 75000;           // 6 MB of 80-byte records
    const int iters = 2000;
    uint4* tab; uint32_t* out;
    hipMalloc(&tab, (size_t)n_rec * 128); hipMalloc(&out, 4);
    hipMemset(tab, 1, (size_t)n_rec * 128);
    const int blocks = 256 * 6;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[5] = {"per-lane record, 5 x 16 B (walk today)", "transposed, stride 80", "transposed, stride 128", "one piece per lane, distinct records", "coalesced 1 KB blocks"};
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            switch (mode) {
                case 0: probe<0><<<blocks, 256>>>(tab, n_rec, iters, out); break;
                case 1: probe<1><<<blocks, 256>>>(tab, n_rec, iters, out); break;
                case 2: probe<2><<<blocks, 256>>>(tab, n_rec, iters, out); break;
                case 3: probe<3><<<blocks, 256>>>(tab, n_rec, iters, out); break;
                default: probe<4><<<blocks, 256>>>(tab, n_rec, iters, out); break;
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) {
                const double wave_rounds = (double)blocks * 4 * iters;
                printf("mode %d  %-42s %8.3f ms   %.1f ns per wave-round of 5 loads, %.2f TB/s\n", mode, names[mode], ms, ms * 1e6 / wave_rounds * (256.0 * 4) /* per SIMD-resident share */ / 1.0, wave_rounds * 5 * 1024 / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
