// probe_gather.hip — what does a wave's scattered record fetch cost on gfx950?  (round 6; the BVH walk's ~41 scattered 16-byte accesses per ray)
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/probe_gather tools/probe_gather.hip && gpurun_out/probe_gather
//
// Every lane of every wave fetches RECORDS at pseudo-random indices of a table (6 MiB ~ C4's tree, 24 MiB ~ C5's, 512 MiB: beyond every cache)
// in several shapes and the program prints vector-memory cycles per wave instruction and per record:
//   own<P, STRIDE>      the lane loads the P 16-byte pieces of its own record (what node8_test / tri_one do: P = 5 @ 80 B, 3 @ 48 B)
//   quad<4>             the four lanes of a quad load the four pieces of ONE 64-byte record per instruction, four instructions serve the quad's
//                       four records (each instruction touches one 64-byte segment per quad instead of four)
// Loads are independent of each other (index = hash(counter)), results are xor-reduced so nothing is dropped; occupancy as the walk's (256
// threads, 7 workgroups per CU).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int P, int STRIDE>
__global__ void __launch_bounds__(256) k_own(const char* __restrict__ tab, uint32_t n_rec, int iters, uint32_t* out) {
    uint32_t acc = 0, ctr = (blockIdx.x * 256u + threadIdx.x) * 7919u;
    for (int it = 0; it < iters; it++) {
        const uint32_t idx = hash32(ctr++) % n_rec;
        const char* base = tab + (size_t)idx * STRIDE;
#pragma unroll
        for (int p = 0; p < P; p++) { const uint4 v = *reinterpret_cast<const uint4*>(base + 16 * p); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// four lanes, four records of 64 bytes: instruction j fetches record j of the quad, lane l its piece l
__global__ void __launch_bounds__(256) k_quad(const char* __restrict__ tab, uint32_t n_rec, int iters, uint32_t* out) {
    uint32_t acc = 0, ctr = (blockIdx.x * 256u + threadIdx.x) * 7919u;
    const uint32_t l = threadIdx.x & 3u;
    for (int it = 0; it < iters; it++) {
        const uint32_t idx = hash32(ctr++) % n_rec;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t idx_j = (uint32_t)__shfl((int)idx, (int)((threadIdx.x & 60u) | (uint32_t)j), 64);      // (a DPP quad broadcast in a real kernel)
            const uint4 v = *reinterpret_cast<const uint4*>(tab + (size_t)idx_j * 64 + 16 * l);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// lane pairs, records of 32 bytes: instruction j fetches record j of the pair, lane l its half
__global__ void __launch_bounds__(256) k_pair32(const char* __restrict__ tab, uint32_t n_rec, int iters, uint32_t* out) {
    uint32_t acc = 0, ctr = (blockIdx.x * 256u + threadIdx.x) * 7919u;
    const uint32_t l = threadIdx.x & 1u;
    for (int it = 0; it < iters; it++) {
        const uint32_t idx = hash32(ctr++) % n_rec;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t idx_j = (uint32_t)__shfl((int)idx, (int)((threadIdx.x & 62u) | (uint32_t)j), 64);
            const uint4 v = *reinterpret_cast<const uint4*>(tab + (size_t)idx_j * 32 + 16 * l);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// 4-byte gathers (the leaf-slot lookup, per-field struct reads)
__global__ void __launch_bounds__(256) k_word(const char* __restrict__ tab, uint32_t n_rec, int iters, uint32_t* out) {
    uint32_t acc = 0, ctr = (blockIdx.x * 256u + threadIdx.x) * 7919u;
    for (int it = 0; it < iters; it++) { const uint32_t idx = hash32(ctr++) % n_rec; acc ^= *reinterpret_cast<const uint32_t*>(tab + (size_t)idx * 4); }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
static float time_ms(F launch, int reps = 5) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const int n_cu = pr.multiProcessorCount;
    const double clk = pr.clockRate * 1e3;      // Hz (nominal)
    printf("%s: %d CUs, %.0f MHz nominal\n", pr.name, n_cu, clk / 1e6);
    uint32_t* out; CHECK(hipMalloc(&out, 64));
    const int iters = 2048, grid = n_cu * 7;
    const double waves = (double)grid * 4.0;
    for (size_t mib : {6, 24, 512}) {
        const size_t bytes = mib << 20;
        char* tab; CHECK(hipMalloc(&tab, bytes + 256)); CHECK(hipMemset(tab, 1, bytes + 256));
        printf("-- table %zu MiB\n", mib);
        auto report = [&](const char* name, float ms, int insts_per_rec) {
            const double cu_cycles = ms * 1e-3 * clk;                                  // per CU: its 28 waves share one vector-memory pipe
            const double recs_per_cu = waves / n_cu * iters * 64.0;
            printf("  %-34s %8.3f ms  %6.2f CU-cycles per record and lane  %6.1f per wave instruction  (%.1f G records/s)\n", name, ms, cu_cycles / recs_per_cu,
                   cu_cycles / recs_per_cu * 64.0 / insts_per_rec, waves * iters * 64.0 / (ms * 1e-3) / 1e9);
        };
        report("own 5 x 16 B @ 80 B (node, now)", time_ms([&] { k_own<5, 80><<<grid, 256>>>(tab, (uint32_t)(bytes / 80), iters, out); }), 5);
        report("own 4 x 16 B @ 64 B (node, 64 B)", time_ms([&] { k_own<4, 64><<<grid, 256>>>(tab, (uint32_t)(bytes / 64), iters, out); }), 4);
        report("own 8 x 16 B @ 128 B", time_ms([&] { k_own<8, 128><<<grid, 256>>>(tab, (uint32_t)(bytes / 128), iters, out); }), 8);
        report("own 3 x 16 B @ 48 B (primitive, now)", time_ms([&] { k_own<3, 48><<<grid, 256>>>(tab, (uint32_t)(bytes / 48), iters, out); }), 3);
        report("own 2 x 16 B @ 32 B", time_ms([&] { k_own<2, 32><<<grid, 256>>>(tab, (uint32_t)(bytes / 32), iters, out); }), 2);
        report("own 1 x 16 B @ 16 B", time_ms([&] { k_own<1, 16><<<grid, 256>>>(tab, (uint32_t)(bytes / 16), iters, out); }), 1);
        report("own 1 x 4 B", time_ms([&] { k_word<<<grid, 256>>>(tab, (uint32_t)(bytes / 4), iters, out); }), 1);
        report("quad: 4 records of 64 B, 4 insts", time_ms([&] { k_quad<<<grid, 256>>>(tab, (uint32_t)(bytes / 64), iters, out); }), 4);
        report("pair: 2 records of 32 B, 2 insts", time_ms([&] { k_pair32<<<grid, 256>>>(tab, (uint32_t)(bytes / 32), iters, out); }), 2);
        CHECK(hipFree(tab));
    }
    return 0;
}
