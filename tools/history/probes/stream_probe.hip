// stream_probe.hip - what the memory system gives a kernel with k_shade's access pattern and no arithmetic:
// 16 Mi entries in 32 sub-queues, a persistent grid of 4 workgroups per CU, every tile row reads R 4-byte SoA streams and writes W.
//   mode 0: SoA, 4 bytes per lane per stream (the queues as they are): R = 13 reads, W = 22 writes
//   mode 1: the same bytes as 16-byte pieces of per-entry records (AoS: 3 + 1/4 reads, 5 + 1/2 writes rounded to 4 / 6 float4 per entry)
//   mode 2: SoA, reads only;  mode 3: SoA, writes only
// then the read side alone by load width and workgroups per CU.
// Measured (MI355X): mode 0 5.13 TB/s, mode 1 3.38, mode 2 5.53, mode 3 5.57; 12 read streams 5.8-6.1 TB/s whatever the width (4 / 8 / 16 bytes per lane) or the
// occupancy (4 / 8 / 16 workgroups per CU) - as long as a wave's loads are independent (the first version of this probe accumulated in a
// run-time loop, one load in flight per wave: 2.3 TB/s).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o tools/probes/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define NQ 32
template <int MODE>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ in, float* __restrict__ out, uint32_t cap, uint32_t subcap, int R, int W) {
    const uint32_t q = blockIdx.x % NQ, first = (blockIdx.x / NQ) * 256u, stride = (gridDim.x / NQ) * 256u;
    for (uint32_t base = first; base < subcap; base += stride) {
        const uint32_t e = q * subcap + base + threadIdx.x;
        if (MODE == 1) {
            const float4* i4 = reinterpret_cast<const float4*>(in); float4* o4 = reinterpret_cast<float4*>(out);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < 4; k++) { const float4 v = i4[(size_t)e * 4 + k]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            for (int k = 0; k < 6; k++) o4[(size_t)e * 6 + k] = make_float4(acc.x + k, acc.y, acc.z, acc.w);
        } else {
            float acc = 0.f; float v[13];
            if (MODE != 3) {
#pragma unroll
                for (int k = 0; k < 13; k++) v[k] = in[(size_t)k * cap + e];          // independent loads, all in flight together
#pragma unroll
                for (int k = 0; k < 13; k++) acc += v[k];
            }
            if (MODE != 2) {
#pragma unroll
                for (int k = 0; k < 22; k++) out[(size_t)k * cap + e] = acc + (float)k;
            }
            if (MODE == 2 && acc == 12345.f) out[e] = acc;
        }
    }
}
template <typename T>
__global__ void __launch_bounds__(256) reads(const float* __restrict__ in, float* __restrict__ out, uint32_t cap, uint32_t subcap) {
    constexpr uint32_t W = sizeof(T) / 4;
    const uint32_t q = blockIdx.x % NQ, first = (blockIdx.x / NQ) * 256u * W, stride = (gridDim.x / NQ) * 256u * W;
    float acc = 0.f;
    for (uint32_t base = first; base < subcap; base += stride) {
        const uint32_t e = q * subcap + base + threadIdx.x * W;
        T v[12];
#pragma unroll
        for (int k = 0; k < 12; k++) v[k] = *reinterpret_cast<const T*>(in + (size_t)k * cap + e);
#pragma unroll
        for (int k = 0; k < 12; k++) acc += reinterpret_cast<const float*>(&v[k])[0] + reinterpret_cast<const float*>(&v[k])[W - 1];
    }
    if (acc == 12345.f) out[threadIdx.x] = acc;
}
int main() {
    const uint32_t subcap = 1u << 19, cap = subcap * NQ;       // 16 Mi entries
    float *in, *out;
    hipMalloc(&in, (size_t)cap * 16 * 4); hipMalloc(&out, (size_t)cap * 24 * 4);
    hipMemset(in, 0, (size_t)cap * 16 * 4); hipMemset(out, 0, (size_t)cap * 24 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * 4;
    const char* names[4] = {"SoA 13 reads + 22 writes", "AoS 4 + 6 float4", "SoA 13 reads", "SoA 22 writes"};
    const double bytes[4] = {35.0 * 4, 40.0 * 4, 13.0 * 4, 22.0 * 4};
    for (int mode = 0; mode < 4; mode++) for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        if (mode == 0) probe<0><<<grid, 256>>>(in, out, cap, subcap, 13, 22);
        if (mode == 1) probe<1><<<grid, 256>>>(in, out, cap, subcap, 13, 22);
        if (mode == 2) probe<2><<<grid, 256>>>(in, out, cap, subcap, 13, 22);
        if (mode == 3) probe<3><<<grid, 256>>>(in, out, cap, subcap, 13, 22);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep == 2) printf("mode %d  %-28s %7.3f ms  %.2f TB/s\n", mode, names[mode], ms, bytes[mode] * cap / (ms * 1e-3) / 1e12);
    }
    // the read side alone, by bytes per lane per load and workgroups per CU: what sets the 2.3 TB/s of mode 2
    for (int wg = 4; wg <= 16; wg *= 2) for (int width = 1; width <= 4; width *= 2) {
        float ms = 0.f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            if (width == 1) reads<float><<<256 * wg, 256>>>(in, out, cap, subcap);
            if (width == 2) reads<float2><<<256 * wg, 256>>>(in, out, cap, subcap);
            if (width == 4) reads<float4><<<256 * wg, 256>>>(in, out, cap, subcap);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        }
        printf("reads: 12 streams, %2d bytes per lane per load, %2d workgroups per CU  %7.3f ms  %.2f TB/s\n", 4 * width, wg, ms, 12.0 * 4 * cap / (ms * 1e-3) / 1e12);
    }
    return 0;
}
