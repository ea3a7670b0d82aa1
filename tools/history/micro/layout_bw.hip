// Micro-benchmark: the shade stage's memory pattern (read a 16-word record, write a 12-word and an 11-word record, both
// through ballot-compacted appends) in two layouts: SoA arrays with stride = capacity (what the queues use) vs AoSoA blocks
// (64 entries x W words, 256-byte rows).  hipcc --offload-arch=gfx950 -O3 tools/micro/layout_bw.hip -o build_exp/layout_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <typename T> __device__ __forceinline__ T ldq(const T* base, uint32_t off) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off); }
template <typename T> __device__ __forceinline__ void stq(T* base, uint32_t off, T v) { *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off) = v; }
__device__ __forceinline__ uint32_t lane() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t append(bool f, uint32_t* c) {
    unsigned long long m = __ballot(f); uint32_t b = 0;
    if (lane() == 0 && m) b = atomicAdd(c, (uint32_t)__popcll(m));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)b) + (uint32_t)__popcll(m & ((1ull << lane()) - 1ull));
}
// MODE 0: SoA, word w of slot s at base + (w * cap + s) * 4.   MODE 1: AoSoA, at base + ((s >> 6) * W + w) * 256 + (s & 63) * 4.
template <int MODE, int W> __device__ __forceinline__ uint32_t slot_off(uint32_t s) { return MODE == 0 ? (s << 2) : ((s >> 6) * (W * 256u) + ((s & 63u) << 2)); }
template <int MODE> __device__ __forceinline__ const float* word_base(const float* b, uint32_t cap, int w) { return MODE == 0 ? b + (size_t)w * cap : b + w * 64; }
template <int MODE> __device__ __forceinline__ float* word_base(float* b, uint32_t cap, int w) { return MODE == 0 ? b + (size_t)w * cap : b + w * 64; }
template <int MODE>
__global__ void __launch_bounds__(256) k(const float* in, float* outA, float* outB, uint32_t cap, uint32_t n_per_q, uint32_t subcap, int nq, uint32_t* cntA, uint32_t* cntB, float keepA, float keepB) {
    const int q = blockIdx.x % nq;
    const uint32_t first = (blockIdx.x / nq) * 256, stride = (gridDim.x / nq) * 256, qb = q * subcap;
    for (uint32_t base = first; base < n_per_q; base += stride) {
        const uint32_t pos = base + threadIdx.x; const bool ok = pos < n_per_q;
        const uint32_t io = slot_off<MODE, 16>(qb + (ok ? pos : n_per_q - 1));
        float v[16];
#pragma unroll
        for (int w = 0; w < 16; w++) v[w] = ldq(word_base<MODE>(in, cap, w), io);
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 16; w++) acc += v[w];
        const bool a = ok && (v[0] < keepA), b = ok && (v[1] < keepB);
        const uint32_t pa = append(a, &cntA[q * 32]);
        if (a) { const uint32_t so = slot_off<MODE, 12>(qb + pa);
#pragma unroll
            for (int w = 0; w < 12; w++) stq(word_base<MODE>(outA, cap, w), so, v[w] + acc); }
        const uint32_t pb = append(b, &cntB[q * 32]);
        if (b) { const uint32_t so = slot_off<MODE, 11>(qb + pb);
#pragma unroll
            for (int w = 0; w < 11; w++) stq(word_base<MODE>(outB, cap, w), so, v[w] - acc); }
    }
}
int main() {
    const int nq = 32; const uint32_t subcap = 262144, cap = subcap * nq;       // 8 Mi entries, as the C2 batch
    float *in, *oa, *ob; uint32_t *ca, *cb;
    CK(hipMalloc(&in, (size_t)cap * 16 * 4)); CK(hipMalloc(&oa, (size_t)cap * 12 * 4)); CK(hipMalloc(&ob, (size_t)cap * 12 * 4));
    CK(hipMalloc(&ca, nq * 128)); CK(hipMalloc(&cb, nq * 128));
    std::vector<float> h((size_t)cap * 16);
    uint32_t x = 12345u; for (auto& f : h) { x = x * 1664525u + 1013904223u; f = (float)(x >> 8) * (1.0f / 16777216.0f); }
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) for (int blocks_per_cu : {4, 8}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemset(ca, 0, nq * 128)); CK(hipMemset(cb, 0, nq * 128));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, in, oa, ob, cap, subcap, subcap, nq, ca, cb, 0.78f, 0.69f);
            else hipLaunchKernelGGL(k<1>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, in, oa, ob, cap, subcap, subcap, nq, ca, cb, 0.78f, 0.69f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double bytes = (double)cap * (64 + 0.78 * 48 + 0.69 * 44);
        printf("%s  %d blocks/CU: %.3f ms for %u entries -> %.0f GB/s algorithmic\n", mode ? "AoSoA" : "SoA  ", blocks_per_cu, best, cap, bytes / best / 1e6);
    }
    return 0;
}
