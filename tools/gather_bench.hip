// gather_bench.hip — microbenchmark: random-gather ceiling of MI355X HBM for the access
// shapes the anchor kernel can use.  Standalone (hipcc --offload-arch=gfx950 -O3).
//   mode 0: quad-cooperative 64-B bucket  (4 lanes x 16 B)        <- k_anchor's shape
//   mode 1: 8 lanes x 16 B = one 128-B line per probe
//   mode 2: one lane per 64-B bucket (4 x dwordx4 per lane)
//   mode 3: 2 lanes x 16 B = 32-B sector per probe
// Prints G probes/s and GB/s for each (table size, unroll).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}

template <int LANES, int U>   // LANES lanes cooperate on one probe of LANES*16 bytes
__global__ __launch_bounds__(256) void k_gather(const uint8_t *tbl, uint64_t nunits, uint64_t probes_per_group, uint32_t *sink) {
    const uint64_t gid = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LANES;   // probe group id
    const int j = threadIdx.x % LANES;
    uint32_t acc = 0;
    for (uint64_t i = 0; i < probes_per_group; i += U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t h = mix(gid * 0x9E3779B97F4A7C15ull + i + u);
            uint64_t b = __umul64hi(h, nunits);
            v[u] = *reinterpret_cast<const uint4 *>(tbl + b * (LANES * 16) + j * 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int U>  // one lane reads a whole 64-B bucket with 4 loads
__global__ __launch_bounds__(256) void k_gather_lane64(const uint8_t *tbl, uint64_t nunits, uint64_t probes_per_group, uint32_t *sink) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (uint64_t i = 0; i < probes_per_group; i += U) {
        uint4 v[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint64_t h = mix(gid * 0x9E3779B97F4A7C15ull + i + u);
            uint64_t b = __umul64hi(h, nunits);
            const uint4 *p = reinterpret_cast<const uint4 *>(tbl + b * 64);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[u][c] = p[c];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc ^= v[u][c].x ^ v[u][c].y ^ v[u][c].z ^ v[u][c].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F>
double time_ms(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char **argv) {
    double gbs[] = {0.25, 6.0, 24.0, 96.0};
    uint32_t *sink; CK(hipMalloc(&sink, 4));
    for (double gb : gbs) {
        uint64_t bytes = (uint64_t)(gb * (1ull << 30)) & ~127ull;
        uint8_t *tbl; if (hipMalloc(&tbl, bytes) != hipSuccess) { printf("skip %.1f GB\n", gb); continue; }
        CK(hipMemset(tbl, 1, bytes));
        const int wgs = 256 * 8;          // 8 WGs per CU
        const uint64_t threads = (uint64_t)wgs * 256;
#define RUN(NAME, KERN, LANES_, UNITB, PPG)                                                       \
        { uint64_t nunits = bytes / (UNITB); uint64_t ppg = (PPG);                                    \
          double ms = time_ms([&]() { hipLaunchKernelGGL(KERN, dim3(wgs), dim3(256), 0, 0, tbl, nunits, ppg, sink); }, 3); \
          double probes = (double)(threads / (LANES_)) * ppg;                                         \
          printf("%-28s table %5.2f GB : %7.2f G probes/s  %7.1f GB/s\n", NAME, gb, probes / ms / 1e6, probes * (UNITB) / ms / 1e6); }
        RUN("quad64  U=4", (k_gather<4, 4>), 4, 64, 256)
        RUN("quad64  U=8", (k_gather<4, 8>), 4, 64, 256)
        RUN("quad64  U=16", (k_gather<4, 16>), 4, 64, 256)
        RUN("oct128  U=8", (k_gather<8, 8>), 8, 128, 256)
        RUN("oct128  U=16", (k_gather<8, 16>), 8, 128, 256)
        RUN("pair32  U=8", (k_gather<2, 8>), 2, 32, 256)
        RUN("pair32  U=16", (k_gather<2, 16>), 2, 32, 256)
        RUN("lane64  U=2", (k_gather_lane64<2>), 1, 64, 64)
        RUN("lane64  U=4", (k_gather_lane64<4>), 1, 64, 64)
        CK(hipFree(tbl));
    }
    return 0;
}
