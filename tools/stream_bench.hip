// Streaming-read rates of the access geometries the statistics kernels use (gfx950):
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_bench tools/stream_bench.hip && tools/stream_bench
// A: grid-stride, 16 B per lane, fully coalesced (what the memory system can do)
// B: 128-thread workgroups walk contiguous ranges (RANGE bytes each), lane t reads 4 x 8 B at
//    32 t + 8 j per 4 KB step (k_epilogue<1> on 8-byte rows)
// C: same ranges, lane t reads 8 B at 8 t + 1024 j (rows t, t + 128, ...: coalesced per instruction)
// D: same ranges, 16 B per lane coalesced, 2 KB per step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void kA(const uint4 *p, uint64_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345) out[0] = acc;
}
template <int MODE>
__global__ __launch_bounds__(128) void kR(const uint8_t *p, uint64_t range, uint32_t *out) {
    const uint8_t *b = p + (uint64_t)blockIdx.x * range;
    uint32_t acc = 0;
    const int t = threadIdx.x;
    for (uint64_t off = 0; off < range; off += 4096) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint2 v = *reinterpret_cast<const uint2 *>(b + off + 32 * t + 8 * j);
                acc += __popc(v.x) + __popc(v.y);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint2 v = *reinterpret_cast<const uint2 *>(b + off + 8 * t + 1024 * j);
                acc += __popc(v.x) + __popc(v.y);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint4 v = *reinterpret_cast<const uint4 *>(b + off + 16 * t + 2048 * j);
                acc += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
            }
        } else if (MODE == 3) {  // E: 4 B per lane, 512 B per workgroup and load, 8 loads (k_epilogue_words at 16-byte rows)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += __popc(*reinterpret_cast<const uint32_t *>(b + off + 4 * t + 512 * j));
        }
    }
    if (acc == 0x12345) out[0] = acc;
}
// deeper pipelines over the same ranges: NL loads of VB bytes per lane issued per step, DEPTH steps in flight
template <int VB, int NL, int DEPTH>
__global__ __launch_bounds__(128) void kP(const uint8_t *p, uint64_t range, uint32_t *out) {
    const uint8_t *b = p + (uint64_t)blockIdx.x * range;
    const int t = threadIdx.x;
    constexpr uint64_t STEP = (uint64_t)128 * VB * NL;
    constexpr int NW = VB / 4;
    uint32_t buf[DEPTH][NL][NW];
    uint32_t acc = 0;
    auto issue = [&](uint64_t off, uint32_t (&o)[NL][NW]) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const uint8_t *q = b + off + (uint64_t)VB * t + (uint64_t)128 * VB * j;
            if (NW == 1) o[j][0] = *reinterpret_cast<const uint32_t *>(q);
            else if (NW == 2) { const uint2 v = *reinterpret_cast<const uint2 *>(q); o[j][0] = v.x; o[j][1] = v.y; }
            else { const uint4 v = *reinterpret_cast<const uint4 *>(q); o[j][0] = v.x; o[j][1] = v.y; o[j][2] = v.z; o[j][3] = v.w; }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d * STEP, buf[d]);
    for (uint64_t off = 0; off < range; off += STEP * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint64_t nxt = off + (uint64_t)(d + DEPTH - 1) * STEP;
            issue(nxt < range ? nxt : 0, buf[(d + DEPTH - 1) % DEPTH]);
#pragma unroll
            for (int j = 0; j < NL; ++j)
#pragma unroll
                for (int w = 0; w < NW; ++w) acc += __popc(buf[d][j][w]);
        }
    }
    if (acc == 0x12345) out[0] = acc;
}

// the one-byte statistics pass's geometry: a PERSISTENT grid of 128-thread workgroups over contiguous ranges of a 0.8 GB
// buffer, lane t reads 2 x 16 B at 32 t per 4 KB step, DEPTH - 1 steps ahead in registers; the grid sets the resident
// waves (3072 workgroups = 6 waves per SIMD, k_epilogue<0>'s occupancy)
template <int DEPTH>
__global__ __launch_bounds__(128) void kE(const uint8_t *p, uint64_t bytes, uint32_t *out) {
    const uint64_t steps = bytes / 4096;
    const uint64_t s0 = steps * blockIdx.x / gridDim.x, s1 = steps * (blockIdx.x + 1) / gridDim.x;
    const int t = threadIdx.x;
    uint4 buf[DEPTH][2];
    uint32_t acc = 0;
    auto issue = [&](uint64_t st, uint4 (&o)[2]) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p + st * 4096 + 32 * t);
        o[0] = q[0];
        o[1] = q[1];
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(s0 + d < s1 ? s0 + d : s0, buf[d]);
    for (uint64_t st = s0; st < s1; st += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint64_t nxt = st + d + DEPTH - 1;
            issue(nxt < s1 ? nxt : s0, buf[(d + DEPTH - 1) % DEPTH]);
            const uint4 x = buf[d][0], y = buf[d][1];
            acc += __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w) + __popc(y.x) + __popc(y.y) + __popc(y.z) + __popc(y.w);
        }
    }
    if (acc == 0x12345) out[0] = acc;
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'e') {
        const uint64_t bytes = 800ull << 20;
        uint8_t *d;
        uint32_t *o;
        hipMalloc(&d, bytes);
        hipMalloc(&o, 4);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        for (int dirty = 0; dirty < 2; ++dirty)
            for (unsigned grid : {1024u, 2048u, 3072u, 4096u, 8192u}) {
                float ms[3] = {0, 0, 0};
                for (int rep = 0; rep < 6; ++rep) {
#define RUNE(D, SLOT)                                                                         \
    if (dirty) hipMemsetAsync(d, rep + 1, bytes, 0);                                          \
    hipEventRecord(a);                                                                        \
    hipLaunchKernelGGL(kE<D>, dim3(grid), dim3(128), 0, 0, d, bytes, o);                      \
    hipEventRecord(b);                                                                        \
    hipEventSynchronize(b);                                                                   \
    { float m; hipEventElapsedTime(&m, a, b); if (rep) ms[SLOT] += m / 5; }
                    RUNE(2, 0) RUNE(3, 1) RUNE(4, 2)
                }
                printf("E %s grid %5u (%.1f waves/SIMD)  1 ahead %.3f ms %.2f TB/s | 2 ahead %.3f ms %.2f TB/s | 3 ahead %.3f ms %.2f TB/s\n",
                       dirty ? "just written" : "read before ", grid, grid * 2 / 1024.0, ms[0], bytes / ms[0] / 1e9, ms[1], bytes / ms[1] / 1e9,
                       ms[2], bytes / ms[2] / 1e9);
            }
        return 0;
    }

    const uint64_t bytes = 10ull << 30;
    uint8_t *d;
    uint32_t *o;
    hipMalloc(&d, bytes);
    hipMalloc(&o, 4);
    hipMemset(d, 1, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto report = [&](const char *name, float ms) { printf("%-40s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12); };
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(a);
        hipLaunchKernelGGL(kA, dim3(256 * 8), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(d), bytes / 16, o);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        report("A grid-stride 16 B/lane", ms);
        for (uint64_t range : {524288ull, 1048576ull, 65536ull}) {
            char nm[64];
            const unsigned grid = (unsigned)(bytes / range);
#define RUNR(M, LABEL)                                                                        \
    hipEventRecord(a);                                                                        \
    hipLaunchKernelGGL(kR<M>, dim3(grid), dim3(128), 0, 0, d, range, o);                      \
    hipEventRecord(b);                                                                        \
    hipEventSynchronize(b);                                                                   \
    hipEventElapsedTime(&ms, a, b);                                                           \
    snprintf(nm, sizeof nm, "%s range %llu KB", LABEL, (unsigned long long)(range >> 10));    \
    report(nm, ms);
            RUNR(0, "B 4x8B at stride 32") RUNR(1, "C 8B coalesced x4") RUNR(2, "D 16B coalesced x2") RUNR(3, "E 4B coalesced x8")
#define RUNP(VB, NL, DEPTH)                                                                   \
    hipEventRecord(a);                                                                        \
    hipLaunchKernelGGL((kP<VB, NL, DEPTH>), dim3(grid), dim3(128), 0, 0, d, range, o);        \
    hipEventRecord(b);                                                                        \
    hipEventSynchronize(b);                                                                   \
    hipEventElapsedTime(&ms, a, b);                                                           \
    snprintf(nm, sizeof nm, "P %d B x %d loads, %d steps ahead, %llu KB", VB, NL, DEPTH - 1, (unsigned long long)(range >> 10)); \
    report(nm, ms);
            RUNP(4, 8, 2) RUNP(4, 16, 2) RUNP(4, 8, 4) RUNP(8, 8, 2) RUNP(16, 2, 2) RUNP(16, 4, 2) RUNP(16, 8, 2) RUNP(16, 4, 4) RUNP(16, 2, 4)
        }
    }
    return 0;
}
