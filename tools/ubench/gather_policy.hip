// What does ONE random 8-byte load cost in HBM traffic on gfx950, and does the cache policy of the load change it?
// The discover epilogue gathers targets[database index] per hit and the compare kernel's flush gathers tidx[slot]: PMC FETCH_SIZE says
// ~128 B per 8-byte gather (profiles/r02/pmc_other_kernels.txt).  This microbenchmark issues the same gather (1.2e7 random indices
// into a 2.4 GB table) with every scope / non-temporal combination the ISA offers; run it under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./gather_policy
// to read the bytes per gather of each variant beside its time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill(uint64_t *t, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) t[i] = i * 0x9E3779B97F4A7C15ull;
}
__global__ void k_index(uint32_t *idx, uint32_t m, uint64_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    idx[i] = (uint32_t)(z % n);
}

#define GATHER(NAME, MODS)                                                                                             \
    __global__ __launch_bounds__(256) void NAME(const uint64_t *__restrict__ t, const uint32_t *__restrict__ idx, uint32_t m, uint64_t *__restrict__ out) { \
        const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;                                                       \
        if (i >= m) return;                                                                                             \
        const uint64_t *p = t + idx[i];                                                                                 \
        uint64_t v;                                                                                                     \
        asm volatile("global_load_dwordx2 %0, %1, off " MODS "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");   \
        out[i] = v;                                                                                                     \
    }
GATHER(k_gather_plain, "")
GATHER(k_gather_sc0, "sc0")
GATHER(k_gather_sc1, "sc1")
GATHER(k_gather_sc0_sc1, "sc0 sc1")
GATHER(k_gather_nt, "nt")
GATHER(k_gather_nt_sc0, "sc0 nt")
GATHER(k_gather_nt_sc1, "sc1 nt")
GATHER(k_gather_nt_sc0_sc1, "sc0 sc1 nt")

// the same with a 4-byte element (tidx[slot])
#define GATHER4(NAME, MODS)                                                                                            \
    __global__ __launch_bounds__(256) void NAME(const uint32_t *__restrict__ t, const uint32_t *__restrict__ idx, uint32_t m, uint32_t *__restrict__ out) { \
        const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;                                                       \
        if (i >= m) return;                                                                                             \
        const uint32_t *p = t + idx[i];                                                                                 \
        uint32_t v;                                                                                                     \
        asm volatile("global_load_dword %0, %1, off " MODS "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");     \
        out[i] = v;                                                                                                     \
    }
GATHER4(k_gather4_plain, "")
GATHER4(k_gather4_nt, "nt")
GATHER4(k_gather4_nt_sc0_sc1, "sc0 sc1 nt")

template <typename K, typename T, typename O>
static void run(const char *name, K kern, const T *t, const uint32_t *idx, uint32_t m, O *out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3((m + 255) / 256), dim3(256), 0, 0, t, idx, m, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-24s %.3f ms for %u gathers = %.1f ps each\n", name, best, m, best * 1e9 / m);
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 300000000ull;
    const uint32_t m = argc > 2 ? (uint32_t)strtoul(argv[2], 0, 10) : 11600000u;
    uint64_t *t, *out;
    uint32_t *idx;
    CHECK(hipMalloc(&t, n * 8)); CHECK(hipMalloc(&out, (size_t)m * 8)); CHECK(hipMalloc(&idx, (size_t)m * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, t, n);
    hipLaunchKernelGGL(k_index, dim3((m + 255) / 256), dim3(256), 0, 0, idx, m, n);
    CHECK(hipDeviceSynchronize());
#define RUN(K) run(#K, K, t, idx, m, out);
    RUN(k_gather_plain) RUN(k_gather_sc0) RUN(k_gather_sc1) RUN(k_gather_sc0_sc1) RUN(k_gather_nt) RUN(k_gather_nt_sc0) RUN(k_gather_nt_sc1) RUN(k_gather_nt_sc0_sc1)
#define RUN4(K) run(#K, K, (const uint32_t *)t, idx, m, (uint32_t *)out);
    RUN4(k_gather4_plain) RUN4(k_gather4_nt) RUN4(k_gather4_nt_sc0_sc1)
    return 0;
}
