// VALU issue cost of the instructions k_probe leans on, relative to v_add_u32 (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate
// One kernel per instruction: 8 waves per SIMD, each runs a dependent-free unrolled stream of the
// instruction (inline asm, 8 independent registers), so the SIMD's issue rate is what is measured.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, ASM1)                                                                      \
    __global__ __launch_bounds__(512) void NAME(uint32_t *out, int iters, uint32_t seed) {      \
        uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 9,   \
                 r5 = r0 * 11, r6 = r0 * 13, r7 = r0 * 15;                                      \
        uint64_t mask = 0x5555555555555555ull ^ seed;                                            \
        uint32_t x0 = r0 + 1, x1 = r0 + 2, x2 = r0 + 3, x3 = r0 + 4, x4 = r0 + 5, x5 = r0 + 6, x6 = r0 + 7, x7 = r0 + 8; \
        uint64_t q0 = r0, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7;        \
        for (int i = 0; i < iters; ++i) {                                                       \
            ASM1(0) ASM1(1) ASM1(2) ASM1(3) ASM1(4) ASM1(5) ASM1(6) ASM1(7)                     \
            ASM1(0) ASM1(1) ASM1(2) ASM1(3) ASM1(4) ASM1(5) ASM1(6) ASM1(7)                     \
            ASM1(0) ASM1(1) ASM1(2) ASM1(3) ASM1(4) ASM1(5) ASM1(6) ASM1(7)                     \
            ASM1(0) ASM1(1) ASM1(2) ASM1(3) ASM1(4) ASM1(5) ASM1(6) ASM1(7)                     \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^     \
            (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7 ^ mask) ^ x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;                                   \
    }

#define A_ADD(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r##n) : "v"(seed));
#define A_XOR(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r##n) : "v"(seed));
#define A_MULLO(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r##n) : "v"(seed));
#define A_MULHI(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r##n) : "v"(seed));
#define A_MUL24(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r##n) : "v"(seed));
#define A_ALIGN(n) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r##n) : "v"(seed));
#define A_BFREV(n) asm volatile("v_bfrev_b32 %0, %0" : "+v"(r##n));
#define A_BFI(n) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(r##n) : "v"(seed));
#define A_MIN(n) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r##n) : "v"(seed));
#define A_DPP(n) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r##n));
#define A_DPPROW(n) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r##n));
#define A_SHR64(n) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q##n));
#define A_SHR64V(n) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q##n) : "v"(seed));
#define A_ADD64(n) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q##n) : "v"(q0));
#define A_CMP64(n) asm volatile("v_cmp_ne_u64 vcc, %0, %1" : : "v"(q##n), "v"(q0) : "vcc");
#define A_CMP32(n) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(r##n), "v"(seed) : "vcc");
#define A_CNDM(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##n) : "v"(seed) : "vcc");
#define A_CNDM2(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##n) : "v"(seed));
#define A_CNDMS(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r##n) : "v"(seed), "s"(mask));
#define A_CMPCND(n) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##n) : "v"(seed) : "vcc");
#define A_CMPCNDS(n) asm volatile("v_cmp_ne_u32 %1, %0, %2\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(r##n), "+s"(mask) : "v"(seed));
#define A_CMP64CND(n) asm volatile("v_cmp_ne_u64 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(r##n) : "v"(q##n), "v"(q0), "v"(seed) : "vcc");
#define A_CMPCND2(n) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %1, vcc" : "+v"(r##n), "+v"(seed) , "+v"(x##n): : "vcc");
#define A_CMPCND2S(n) asm volatile("v_cmp_ne_u32 %1, %0, %2\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %2, %1\n v_cndmask_b32_e64 %3, %3, %2, %1" : "+v"(r##n), "+s"(mask), "+v"(seed), "+v"(x##n));
#define A_CND64VCC(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(r##n) : "v"(seed));
#define A_CMP64CND2(n) asm volatile("v_cmp_gt_u64 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %0, %3, vcc\n v_cndmask_b32 %4, %4, %3, vcc" : "+v"(r##n) : "v"(q##n), "v"(q0), "v"(seed), "v"(x##n) : "vcc");
#define A_MAD64(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##n) : "v"(r##n), "v"(seed) : "vcc");
#define A_MBCNT(n) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r##n) : "v"(seed));
#define A_MOV64(n) asm volatile("v_mov_b64 %0, %1" : "=v"(q##n) : "v"(q0));
#define A_PKADD(n) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r##n) : "v"(seed));
#define A_AND_OR(n) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r##n) : "v"(seed));
#define A_XAD(n) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(r##n) : "v"(seed));
#define A_PERM(n) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r##n) : "v"(seed));
#define A_BCNT(n) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r##n) : "v"(seed));
#define A_MINDPP(n) asm volatile("s_nop 1\n v_min_u32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r##n) : "v"(seed));
#define A_MINDPPROW(n) asm volatile("s_nop 1\n v_min_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r##n) : "v"(seed));
#define A_CMP64S(n) asm volatile("v_cmp_ne_u64_e64 %0, %1, %2" : "=s"(mask) : "v"(q##n), "v"(q0));
#define A_CMP32S(n) asm volatile("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(mask) : "v"(r##n), "v"(seed));
#define A_NOT(n) asm volatile("v_not_b32 %0, %0" : "+v"(r##n));
#define A_LSHL(n) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r##n));
#define A_LSHLADD(n) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r##n) : "v"(seed));
#define A_ADD3(n) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r##n) : "v"(seed));
#define A_OR3(n) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(r##n) : "v"(seed));
#define A_AND(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r##n) : "v"(seed));
#define A_BFE(n) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(r##n));
#define A_LSHLADD64(n) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q##n) : "v"(q0));
#define A_SHL64(n) asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(q##n));

KERNEL(k_add, A_ADD) KERNEL(k_xor, A_XOR) KERNEL(k_mullo, A_MULLO) KERNEL(k_mulhi, A_MULHI) KERNEL(k_mul24, A_MUL24)
KERNEL(k_align, A_ALIGN) KERNEL(k_bfrev, A_BFREV) KERNEL(k_bfi, A_BFI) KERNEL(k_min, A_MIN) KERNEL(k_dpp, A_DPP)
KERNEL(k_dpprow, A_DPPROW) KERNEL(k_shr64, A_SHR64) KERNEL(k_shr64v, A_SHR64V) KERNEL(k_add64, A_ADD64)
KERNEL(k_cmp64, A_CMP64) KERNEL(k_cmp32, A_CMP32) KERNEL(k_cndm, A_CNDM) KERNEL(k_mad64, A_MAD64) KERNEL(k_mbcnt, A_MBCNT)
KERNEL(k_mov64, A_MOV64) KERNEL(k_pkadd, A_PKADD) KERNEL(k_andor, A_AND_OR) KERNEL(k_xad, A_XAD) KERNEL(k_perm, A_PERM)
KERNEL(k_mindpp, A_MINDPP) KERNEL(k_mindpprow, A_MINDPPROW) KERNEL(k_cmp64s, A_CMP64S) KERNEL(k_cmp32s, A_CMP32S) KERNEL(k_not, A_NOT)
KERNEL(k_lshl, A_LSHL) KERNEL(k_lshladd, A_LSHLADD) KERNEL(k_add3, A_ADD3) KERNEL(k_or3, A_OR3) KERNEL(k_and, A_AND) KERNEL(k_bfe, A_BFE)
KERNEL(k_lshladd64, A_LSHLADD64) KERNEL(k_shl64, A_SHL64)
KERNEL(k_cmpcnd2, A_CMPCND2) KERNEL(k_cmpcnd2s, A_CMPCND2S) KERNEL(k_cnd64vcc, A_CND64VCC) KERNEL(k_cmp64cnd2, A_CMP64CND2)
KERNEL(k_bcnt, A_BCNT) KERNEL(k_cndm2, A_CNDM2) KERNEL(k_cndms, A_CNDMS) KERNEL(k_cmpcnd, A_CMPCND) KERNEL(k_cmpcnds, A_CMPCNDS) KERNEL(k_cmp64cnd, A_CMP64CND)

template <typename K>
static double run(K kern, uint32_t *d_out, int blocks) {
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d_out, 10, 1u);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d_out, iters, 1u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD: (blocks * 8 waves / (CUs * 4 SIMDs)) * iters * 32
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 4;  // 4 x 512 threads = 32 waves per CU = 8 per SIMD
    uint32_t *d_out;
    hipMalloc(&d_out, (size_t)blocks * 512 * 4);
    const double wave_instr_per_simd = 8.0 * 2000 * 32;
    const double clk = p.clockRate * 1e3;  // Hz
    printf("%d CUs, %.0f MHz\n", cus, clk / 1e6);
#define RUN(K) { double ms = run(K, d_out, blocks); printf("%-10s %8.3f ms  %6.2f clk per wave instruction\n", #K, ms, ms * 1e-3 * clk / wave_instr_per_simd); }
    RUN(k_add) RUN(k_xor) RUN(k_mullo) RUN(k_mulhi) RUN(k_mul24) RUN(k_align) RUN(k_bfrev) RUN(k_bfi) RUN(k_min) RUN(k_dpp)
    RUN(k_dpprow) RUN(k_shr64) RUN(k_shr64v) RUN(k_add64) RUN(k_cmp64) RUN(k_cmp32) RUN(k_cndm) RUN(k_mad64) RUN(k_mbcnt)
    RUN(k_cndm2) RUN(k_cndms) RUN(k_cmpcnd) RUN(k_cmpcnds) RUN(k_cmp64cnd) RUN(k_add) RUN(k_xor)
    RUN(k_mov64) RUN(k_pkadd) RUN(k_andor) RUN(k_xad) RUN(k_perm) RUN(k_bcnt)
    RUN(k_mindpp) RUN(k_mindpprow) RUN(k_cmp64s) RUN(k_cmp32s) RUN(k_not) RUN(k_lshl) RUN(k_lshladd) RUN(k_add3) RUN(k_or3) RUN(k_and)
    RUN(k_bfe) RUN(k_lshladd64) RUN(k_shl64)
    // (several instructions per asm statement: divide by their number)
    RUN(k_cmpcnd2) RUN(k_cmpcnd2s) RUN(k_cnd64vcc) RUN(k_cmp64cnd2)
    return 0;
}
