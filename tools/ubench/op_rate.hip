// Issue cost of single integer VALU opcodes on gfx950: 16 independent instances of ONE instruction per loop trip, 8 waves per SIMD,
// every SIMD of the chip busy.  Prints cycles per wave-instruction per SIMD (2 = a SIMD-32 op, 4 = half rate).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

#define DEFKERNEL(NAME, ASMSTR)                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t iters, uint32_t seed) {         \
        uint32_t a[16];                                                                                  \
        const uint32_t b = threadIdx.x * 2654435761u + seed, c = threadIdx.x ^ 0x5bd1e995u;              \
        for (int k = 0; k < 16; ++k) a[k] = b + k * 7919u;                                               \
        for (uint32_t i = 0; i < iters; ++i) {                                                           \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) asm volatile(ASMSTR : "+v"(a[k]) : "v"(b), "v"(c), "s"(seed)); \
        }                                                                                                \
        uint32_t s = 0;                                                                                  \
        for (int k = 0; k < 16; ++k) s += a[k];                                                          \
        if (s == 0x12345) out[0] = s;                                                                    \
    }

DEFKERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEFKERNEL(k_xor_s, "v_xor_b32 %0, %3, %0")
DEFKERNEL(k_add, "v_add_u32 %0, %0, %1")
DEFKERNEL(k_lshr, "v_lshrrev_b32 %0, 1, %0")
DEFKERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
DEFKERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xde")
DEFKERNEL(k_min, "v_min_u32 %0, %0, %1")
DEFKERNEL(k_min3, "v_min3_u32 %0, %0, %1, %2")
DEFKERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEFKERNEL(k_or_sdwa, "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0")
DEFKERNEL(k_cmp, "v_cmp_le_u32 vcc, %0, %1")
DEFKERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
DEFKERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEFKERNEL(k_sad, "v_sad_u8 %0, %0, %1, %2")
DEFKERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
DEFKERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEFKERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEFKERNEL(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2")
DEFKERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
DEFKERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFKERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEFKERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
DEFKERNEL(k_bfe_i32, "v_bfe_i32 %0, %0, 3, 1")
DEFKERNEL(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 5")
DEFKERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFKERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
DEFKERNEL(k_sub, "v_sub_u32 %0, %0, %1")
DEFKERNEL(k_ashr, "v_ashrrev_i32 %0, 31, %0")
DEFKERNEL(k_and, "v_and_b32 %0, %0, %1")
DEFKERNEL(k_xor_lit, "v_xor_b32 %0, 0x12345678, %0")
DEFKERNEL(k_add_inl, "v_add_u32 %0, 1, %0")
DEFKERNEL(k_ffbl, "v_ffbl_b32 %0, %0")
DEFKERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")

template <typename K>
static void run(const char *name, K kern, uint32_t *out) {
    const uint32_t iters = 1 << 14;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, 0, out, iters, 3u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    printf("%-14s %.3f ms  %.2f cycles per wave-instruction per SIMD (8 waves/SIMD)\n", name, best, best * 1e-3 * 2.4e9 / ((double)iters * 16 * 8));
}

int main() {
    uint32_t *out;
    CHECK(hipMalloc(&out, 64));
#define RUN(K) run(#K, K, out);
    RUN(k_xor) RUN(k_xor_s) RUN(k_add) RUN(k_lshr) RUN(k_bcnt) RUN(k_bitop3) RUN(k_min) RUN(k_min3) RUN(k_and_or) RUN(k_or_sdwa) RUN(k_cmp)
    RUN(k_alignbit) RUN(k_perm) RUN(k_sad) RUN(k_mad24) RUN(k_fma) RUN(k_pk_add_u16) RUN(k_dot4) RUN(k_lshl_or) RUN(k_mov_dpp)
    RUN(k_mul_lo) RUN(k_mul24) RUN(k_bfe_i32) RUN(k_bfe_u32) RUN(k_cndmask) RUN(k_lshl_add) RUN(k_sub) RUN(k_ashr) RUN(k_and) RUN(k_xor_lit) RUN(k_add_inl) RUN(k_ffbl) RUN(k_mbcnt)
    return 0;
}
