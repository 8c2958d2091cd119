// Microbenchmarks behind DESIGN.md's issue-rate figures: how many integer VALU wave-instructions per cycle a gfx950 SIMD sustains
// (is an integer wave64 op 2 or 4 cycles?), and what the pair-test inner loop costs when its candidate comes from an SGPR, from an
// LDS broadcast (8-byte / 4-byte) -- the ceiling the compare kernel is measured against.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// A: pure dependent-free integer VALU: 8 independent xor chains
__global__ __launch_bounds__(256) void k_xor(uint32_t *out, uint32_t iters, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 2654435761u + k + seed;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] ^= (a[(k + 1) & 7] >> 1);  // v_lshrrev + v_xor -> or v_bitop/alignbit; counted from the ISA
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 0x12345) out[0] = s;
}

// B: the pair test with the candidate in scalar registers: xor, bitop3, bcnt, min per candidate; 4 candidates per "group"
__global__ __launch_bounds__(256) void k_pair_sgpr(uint32_t *out, const uint64_t *__restrict__ cand, uint32_t iters, uint32_t n_cand) {
    const uint32_t kh = threadIdx.x * 2654435761u, kl = threadIdx.x * 40503u + blockIdx.x;
    uint32_t best = 64, hits = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        const uint64_t *c = cand + ((i * 8) & (n_cand - 1));  // uniform address -> s_load_dwordx4..x16
        uint32_t b = 64;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t g = c[k];
            const uint32_t y = __builtin_amdgcn_bitop3_b32((uint32_t)g, kh ^ (uint32_t)(g >> 32), kl, 0xde);
            b = min(b, (uint32_t)__popc(y));
        }
        if (__builtin_amdgcn_ballot_w64(b <= 1)) hits++;
        best = min(best, b);
    }
    if (best == 77 || hits == 0xFFFFFFFF) out[0] = best;
}

// C: the same with the candidate as an 8-byte LDS broadcast (what k_compare v6 does)
template <int BYTES>
__global__ __launch_bounds__(256) void k_pair_lds(uint32_t *out, const uint64_t *__restrict__ cand, uint32_t iters) {
    __shared__ uint64_t lds[4][256];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < 4; ++k) lds[wave][k * 64 + lane] = cand[k * 64 + lane];
    __syncthreads();
    const uint32_t kh = threadIdx.x * 2654435761u, kl = threadIdx.x * 40503u + blockIdx.x;
    uint32_t best = 64, hits = 0;
    typedef __attribute__((address_space(3))) const uint64_t lds64;
    typedef __attribute__((address_space(3))) const uint32_t lds32;
    for (uint32_t i = 0; i < iters; ++i) {
        uint32_t b = 64;
        const uint32_t base = (i * 8) & 255u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t y;
            if (BYTES == 8) {
                const uint64_t g = ((lds64 *)&lds[wave][0])[(base + k) & 255];
                y = __builtin_amdgcn_bitop3_b32((uint32_t)g, kh ^ (uint32_t)(g >> 32), kl, 0xde);
            } else {
                const uint32_t g = ((lds32 *)&lds[wave][0])[(base + k) & 255];
                const uint32_t x = g ^ kl;
                if (BYTES == 4) y = (x >> 16) | (x & 0xFFFFu);
                else asm("v_or_b32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(y) : "v"(x));  // BYTES == 5: the fold as one SDWA op
            }
            b = min(b, (uint32_t)__popc(y));
        }
        if (__builtin_amdgcn_ballot_w64(b <= 1)) hits++;
        best = min(best, b);
    }
    if (best == 77 || hits == 0xFFFFFFFF) out[0] = best;
}

template <typename F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    uint32_t *out; uint64_t *cand;
    CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&cand, 4096 * 8));
    uint64_t h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 0x9E3779B97F4A7C15ull * (i + 1);
    CHECK(hipMemcpy(cand, h, sizeof h, hipMemcpyHostToDevice));
    const double clk = 2.4e9, simds = 1024;
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
        const unsigned grid = 256 * wps;       // 256 threads = 4 waves = one per SIMD of a CU
        const uint32_t iters = 1 << 15;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_xor, dim3(grid), dim3(256), 0, 0, out, iters, 1u); });
        printf("waves/SIMD %d  k_xor        %.3f ms  -> %.2f cycles per (wave, 8-op group) per SIMD\n", wps, ms, ms * 1e-3 * clk / ((double)iters * wps));
        ms = time_ms([&] { hipLaunchKernelGGL(k_pair_sgpr, dim3(grid), dim3(256), 0, 0, out, cand, iters, 4096u); });
        printf("waves/SIMD %d  k_pair_sgpr  %.3f ms  -> %.2f cycles per candidate row per SIMD\n", wps, ms, ms * 1e-3 * clk / ((double)iters * 8 * wps));
        ms = time_ms([&] { hipLaunchKernelGGL(k_pair_lds<8>, dim3(grid), dim3(256), 0, 0, out, cand, iters); });
        printf("waves/SIMD %d  k_pair_lds8  %.3f ms  -> %.2f cycles per candidate row per SIMD\n", wps, ms, ms * 1e-3 * clk / ((double)iters * 8 * wps));
        ms = time_ms([&] { hipLaunchKernelGGL(k_pair_lds<4>, dim3(grid), dim3(256), 0, 0, out, cand, iters); });
        printf("waves/SIMD %d  k_pair_lds4  %.3f ms  -> %.2f cycles per candidate row per SIMD\n", wps, ms, ms * 1e-3 * clk / ((double)iters * 8 * wps));
        ms = time_ms([&] { hipLaunchKernelGGL(k_pair_lds<5>, dim3(grid), dim3(256), 0, 0, out, cand, iters); });
        printf("waves/SIMD %d  k_pair_sdwa  %.3f ms  -> %.2f cycles per candidate row per SIMD\n", wps, ms, ms * 1e-3 * clk / ((double)iters * 8 * wps));
    }
    (void)simds;
    return 0;
}
