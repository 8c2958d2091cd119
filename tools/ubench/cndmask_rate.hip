// Issue cost of v_cndmask_b32 forms on gfx950 (same harness as op_rate.hip): the lane-select reads a 64-bit scalar mask.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define DEFKERNEL(NAME, ASMSTR)                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t iters, uint32_t seed, uint64_t mask) { \
        uint32_t a[16];                                                                                  \
        const uint32_t b = threadIdx.x * 2654435761u + seed, c = threadIdx.x ^ 0x5bd1e995u;              \
        for (int k = 0; k < 16; ++k) a[k] = b + k * 7919u;                                               \
        for (uint32_t i = 0; i < iters; ++i) {                                                           \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) asm volatile(ASMSTR : "+v"(a[k]) : "v"(b), "v"(c), "s"(mask)); \
        }                                                                                                \
        uint32_t s = 0;                                                                                  \
        for (int k = 0; k < 16; ++k) s += a[k];                                                          \
        if (s == 0x12345) out[0] = s;                                                                    \
    }

DEFKERNEL(k_cnd_e64, "v_cndmask_b32_e64 %0, %0, %1, %3")
DEFKERNEL(k_cnd_e64_const, "v_cndmask_b32_e64 %0, 0, %0, %3")
DEFKERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEFKERNEL(k_cmp_e64, "v_cmp_ne_u32_e64 %3, %0, %1")
// a compare writing vcc followed by a select on vcc (the usual pair)
DEFKERNEL(k_cmp_cnd, "v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %2, vcc")
// the same selection with arithmetic: mask = (int)(x - y) >> 31; r = (a & ~mask) | (b & mask) = bitop3
DEFKERNEL(k_sub_ashr_bitop, "v_sub_u32 %0, %1, %0\n v_ashrrev_i32 %0, 31, %0\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca")

template <typename K>
static void run(const char *name, K kern, uint32_t *out, int per) {
    const uint32_t iters = 1 << 14;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, 0, out, iters, 3u, 0x5555AAAA3333CCCCull);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    printf("%-18s %.3f ms  %.2f cycles per group of %d instruction(s) per SIMD (8 waves/SIMD)\n", name, best, best * 1e-3 * 2.4e9 / ((double)iters * 16 * 8), per);
}

int main() {
    uint32_t *out;
    CHECK(hipMalloc(&out, 64));
    run("k_xor", k_xor, out, 1);
    run("k_cnd_e64", k_cnd_e64, out, 1);
    run("k_cnd_e64_const", k_cnd_e64_const, out, 1);
    run("k_cmp_e64", k_cmp_e64, out, 1);
    run("k_cmp_cnd", k_cmp_cnd, out, 2);
    run("k_sub_ashr_bitop", k_sub_ashr_bitop, out, 3);
    return 0;
}
