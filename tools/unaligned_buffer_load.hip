// round 6 microtest: do raw buffer loads (buffer_load_dwordx3 / x4, offen, sc1) take byte-aligned offsets on gfx950, as the
// global_* instructions do?  (k_probe's fused statistics read ragged rows back that way)
//   hipcc --offload-arch=gfx950 -O2 -o tools/unaligned_buffer_load tools/unaligned_buffer_load.hip && tools/unaligned_buffer_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__global__ void k(const uint8_t *p, uint32_t *out, uint32_t nb) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0x7ffffff0, 0x00020000);
    const uint32_t off = threadIdx.x * nb;
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
    u32x3 w = __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 16);
    uint32_t x = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 16);
    out[threadIdx.x * 8 + 0] = v.x, out[threadIdx.x * 8 + 1] = v.y, out[threadIdx.x * 8 + 2] = v.z, out[threadIdx.x * 8 + 3] = v.w;
    out[threadIdx.x * 8 + 4] = w.x, out[threadIdx.x * 8 + 5] = w.y, out[threadIdx.x * 8 + 6] = w.z, out[threadIdx.x * 8 + 7] = x;
}
int main() {
    std::vector<uint8_t> h(4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d;
    uint32_t *o;
    hipMalloc(&d, h.size());
    hipMalloc(&o, 64 * 8 * 4);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    int bad = 0;
    for (uint32_t nb : {1u, 2u, 3u, 5u, 7u, 9u, 11u, 13u, 15u, 16u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, nb);
        std::vector<uint32_t> r(64 * 8);
        hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
        for (int t = 0; t < 64; ++t) {
            uint32_t e[4];
            for (int i = 0; i < 4; ++i) memcpy(&e[i], &h[t * nb + 4 * i], 4);
            const bool ok = r[t * 8] == e[0] && r[t * 8 + 1] == e[1] && r[t * 8 + 2] == e[2] && r[t * 8 + 3] == e[3] && r[t * 8 + 4] == e[0] &&
                            r[t * 8 + 5] == e[1] && r[t * 8 + 6] == e[2] && r[t * 8 + 7] == e[0];
            if (!ok) ++bad;
        }
        printf("stride %u: %s\n", nb, bad ? "MISMATCH" : "ok");
    }
    printf(bad ? "unaligned buffer loads: NOT supported as byte-exact\n" : "unaligned buffer loads: byte-exact\n");
    return bad != 0;
}
