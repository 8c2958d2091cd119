// tools/probes/bw_probe.hip -- what a streaming read of u64 keys reaches on this GPU, by the shape of the loads (round 5: every kernel of the
// hit ordering that reads the keys once -- k_sort_hist, k_msd_hist, k_segments -- sat at ~2.4 TB/s).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bw_probe tools/probes/bw_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) what the kernels do: 256 threads, 16 rows of 8-byte loads per thread, one chunk of 4096 keys per block
__global__ __launch_bounds__(256) void k_rows8(const uint64_t *__restrict__ keys, uint64_t n, uint64_t *__restrict__ out) {
    const uint64_t base = (uint64_t)blockIdx.x * 4096;
    uint64_t k[16], acc = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const uint64_t i = base + r * 256 + threadIdx.x; k[r] = i < n ? keys[i] : 0; }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc ^= k[r];
    if (acc == 0x123456789ull) out[0] = acc;
}
// (b) the same keys per block, 16-byte loads (two keys per lane and load)
__global__ __launch_bounds__(256) void k_rows16(const uint64_t *__restrict__ keys, uint64_t n, uint64_t *__restrict__ out) {
    const uint64_t base = (uint64_t)blockIdx.x * 4096;
    ulonglong2 k[8]; uint64_t acc = 0;
    const ulonglong2 *p = (const ulonglong2 *)(keys + base);
#pragma unroll
    for (int r = 0; r < 8; ++r) { const uint64_t i = base + 2 * (r * 256 + threadIdx.x); k[r] = i + 1 < n ? p[r * 256 + threadIdx.x] : ulonglong2{0, 0}; }
#pragma unroll
    for (int r = 0; r < 8; ++r) acc ^= k[r].x ^ k[r].y;
    if (acc == 0x123456789ull) out[0] = acc;
}
// (c) a grid that fits the GPU once, every block striding over the keys, 16-byte loads, UNROLL loads in flight
template <int UNROLL>
__global__ __launch_bounds__(256) void k_stride16(const uint64_t *__restrict__ keys, uint64_t n, uint64_t *__restrict__ out) {
    const ulonglong2 *p = (const ulonglong2 *)keys;
    const uint64_t n2 = n / 2, stride = (uint64_t)gridDim.x * 256;
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride * UNROLL) {
        ulonglong2 k[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) k[u] = i + u * stride < n2 ? p[i + u * stride] : ulonglong2{0, 0};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= k[u].x ^ k[u].y;
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
// (d) (a) with an LDS histogram of 512 digits behind the loads (k_sort_hist itself)
__global__ __launch_bounds__(256) void k_hist8(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ table, uint32_t nblocks) {
    __shared__ uint32_t h[512];
    for (uint32_t d = threadIdx.x; d < 512; d += 256) h[d] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const uint64_t i = base + r * 256 + threadIdx.x; if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> 9) & 511u], 1u); }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < 512; d += 256) table[(uint64_t)d * nblocks + blockIdx.x] = h[d];
}
// (d2) the loads first, then the atomics
__global__ __launch_bounds__(256) void k_hist8_pre(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ table, uint32_t nblocks) {
    __shared__ uint32_t h[512];
    for (uint32_t d = threadIdx.x; d < 512; d += 256) h[d] = 0;
    const uint64_t base = (uint64_t)blockIdx.x * 4096;
    uint64_t k[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const uint64_t i = base + r * 256 + threadIdx.x; k[r] = i < n ? keys[i] : ~0ull; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) if (base + r * 256 + threadIdx.x < n) atomicAdd(&h[(uint32_t)(k[r] >> 9) & 511u], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < 512; d += 256) table[(uint64_t)d * nblocks + blockIdx.x] = h[d];
}
// (d4) MODE 0: the counts not written; 1: written block-major (a block's 512 counts contiguous); 2: digit-major
template <int MODE>
__global__ __launch_bounds__(256) void k_hist8_tab(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ table, uint32_t nblocks) {
    __shared__ uint32_t h[512];
    for (uint32_t d = threadIdx.x; d < 512; d += 256) h[d] = 0;
    const uint64_t base = (uint64_t)blockIdx.x * 4096;
    uint64_t k[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const uint64_t i = base + r * 256 + threadIdx.x; k[r] = i < n ? keys[i] : ~0ull; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) if (base + r * 256 + threadIdx.x < n) atomicAdd(&h[(uint32_t)(k[r] >> 9) & 511u], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < 512; d += 256) {
        if (MODE == 0) { if (h[d] == 0xFFFFFFFFu) table[d] = 1; }
        else if (MODE == 1) table[(uint64_t)blockIdx.x * 512 + d] = h[d];
        else table[(uint64_t)d * nblocks + blockIdx.x] = h[d];
    }
}
// (d3) a histogram per wave (4 x 512 counters), summed at the end
__global__ __launch_bounds__(256) void k_hist8_wave(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ table, uint32_t nblocks) {
    __shared__ uint32_t h[4][512];
    for (uint32_t d = threadIdx.x; d < 2048; d += 256) (&h[0][0])[d] = 0;
    const uint64_t base = (uint64_t)blockIdx.x * 4096;
    uint64_t k[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const uint64_t i = base + r * 256 + threadIdx.x; k[r] = i < n ? keys[i] : ~0ull; }
    __syncthreads();
    uint32_t *mine = h[threadIdx.x >> 6];
#pragma unroll
    for (int r = 0; r < 16; ++r) if (base + r * 256 + threadIdx.x < n) atomicAdd(&mine[(uint32_t)(k[r] >> 9) & 511u], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < 512; d += 256) table[(uint64_t)d * nblocks + blockIdx.x] = h[0][d] + h[1][d] + h[2][d] + h[3][d];
}
// (f) LDS atomics alone: 16 per thread on hashed addresses; RET: the returned value is used (ds_add_rtn_u32)
template <bool RET, uint32_t DIG>
__global__ __launch_bounds__(256) void k_lds_atomics(uint32_t *__restrict__ out, uint32_t rounds) {
    __shared__ uint32_t h[DIG];
    for (uint32_t d = threadIdx.x; d < DIG; d += 256) h[d] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        if (RET) acc += atomicAdd(&h[x & (DIG - 1u)], 1u);
        else atomicAdd(&h[x & (DIG - 1u)], 1u);
    }
    __syncthreads();
    if (acc == 0xFFFFFFF1u || h[threadIdx.x & (DIG - 1u)] == 0xFFFFFFFu) out[0] = acc;
}
// (g) the same counting without atomics: every row's lanes find their peers (same digit) with 9 ballots, the last peer adds the count
// to the wave's own counter (what k_sort_scatter does to rank)
__global__ __launch_bounds__(256) void k_lds_ballots(uint32_t *__restrict__ out, uint32_t rounds) {
    __shared__ uint32_t h[4][512];
    for (uint32_t d = threadIdx.x; d < 2048; d += 256) (&h[0][0])[d] = 0;
    __syncthreads();
    uint32_t *mine = h[threadIdx.x >> 6];
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const uint32_t d = x & 511u;
        uint64_t peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 9; ++b) { const uint64_t bal = __ballot((d >> b) & 1); peers &= ((d >> b) & 1) ? bal : ~bal; }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        const uint32_t cnt = (uint32_t)__popcll(peers);
        const uint32_t before = mine[d];
        acc += before + rank;
        __builtin_amdgcn_wave_barrier();
        if (rank == cnt - 1u) mine[d] = before + cnt;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (acc == 0xFFFFFFF1u) out[0] = acc;
}
// (e) copy, 16 bytes per lane, grid-stride
__global__ __launch_bounds__(256) void k_copy16(const uint64_t *__restrict__ keys, uint64_t *__restrict__ dst, uint64_t n) {
    const ulonglong2 *p = (const ulonglong2 *)keys; ulonglong2 *q = (ulonglong2 *)dst;
    const uint64_t n2 = n / 2, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride * 4) {
        ulonglong2 k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * stride < n2) k[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * stride < n2) q[i + u * stride] = k[u];
    }
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 47000000ull;
    uint64_t *keys, *dst, *out; uint32_t *table;
    const uint32_t nb = (uint32_t)((n + 4095) / 4096);
    CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&dst, n * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&table, (size_t)512 * nb * 4));
    std::vector<uint64_t> h(n);
    uint64_t x = 88172645463325252ull;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x; }
    CK(hipMemcpy(keys, h.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, double bytes, auto &&launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0, 0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
    };
    const double B = (double)n * 8;
    time("rows of 8 B, 4096 keys per block", B, [&] { hipLaunchKernelGGL(k_rows8, dim3(nb), dim3(256), 0, 0, keys, n, out); });
    time("rows of 16 B, 4096 keys per block", B, [&] { hipLaunchKernelGGL(k_rows16, dim3(nb), dim3(256), 0, 0, keys, n, out); });
    time("grid-stride 16 B x4, 2048 blocks", B, [&] { hipLaunchKernelGGL(k_stride16<4>, dim3(2048), dim3(256), 0, 0, keys, n, out); });
    time("grid-stride 16 B x8, 2048 blocks", B, [&] { hipLaunchKernelGGL(k_stride16<8>, dim3(2048), dim3(256), 0, 0, keys, n, out); });
    time("grid-stride 16 B x8, 4096 blocks", B, [&] { hipLaunchKernelGGL(k_stride16<8>, dim3(4096), dim3(256), 0, 0, keys, n, out); });
    time("rows of 8 B + LDS histogram + table", B, [&] { hipLaunchKernelGGL(k_hist8, dim3(nb), dim3(256), 0, 0, keys, n, table, nb); });
    time("loads first, then the LDS histogram", B, [&] { hipLaunchKernelGGL(k_hist8_pre, dim3(nb), dim3(256), 0, 0, keys, n, table, nb); });
    time("... counts not written", B, [&] { hipLaunchKernelGGL(k_hist8_tab<0>, dim3(nb), dim3(256), 0, 0, keys, n, table, nb); });
    time("... counts written block-major", B, [&] { hipLaunchKernelGGL(k_hist8_tab<1>, dim3(nb), dim3(256), 0, 0, keys, n, table, nb); });
    time("... counts written digit-major", B, [&] { hipLaunchKernelGGL(k_hist8_tab<2>, dim3(nb), dim3(256), 0, 0, keys, n, table, nb); });
    time("loads first, a histogram per wave", B, [&] { hipLaunchKernelGGL(k_hist8_wave, dim3(nb), dim3(256), 0, 0, keys, n, table, nb); });
    {   // LDS atomics alone: nb blocks x 256 threads x 16 = the same number of updates as the histogram of n keys
        const double U = (double)nb * 256 * 16;
        auto rate = [&](const char *name, auto &&launch) {
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 20; ++i) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / 20;
            printf("%-44s %8.1f us  %6.2f updates/clk/CU (2.4 GHz, 256 CUs)\n", name, us, U / (us * 1e-6) / 2.4e9 / 256);
        };
        rate("LDS atomics alone, 512 counters, no return", [&] { hipLaunchKernelGGL((k_lds_atomics<false, 512>), dim3(nb), dim3(256), 0, 0, (uint32_t *)out, 16u); });
        rate("LDS atomics alone, 512 counters, returning", [&] { hipLaunchKernelGGL((k_lds_atomics<true, 512>), dim3(nb), dim3(256), 0, 0, (uint32_t *)out, 16u); });
        rate("LDS atomics alone, 2048 counters, returning", [&] { hipLaunchKernelGGL((k_lds_atomics<true, 2048>), dim3(nb), dim3(256), 0, 0, (uint32_t *)out, 16u); });
        rate("LDS atomics alone, 16 counters, returning", [&] { hipLaunchKernelGGL((k_lds_atomics<true, 16>), dim3(nb), dim3(256), 0, 0, (uint32_t *)out, 16u); });
        rate("9 ballots + the wave's own counters", [&] { hipLaunchKernelGGL(k_lds_ballots, dim3(nb), dim3(256), 0, 0, (uint32_t *)out, 16u); });
    }
    time("copy 16 B x4 (read + write)", 2 * B, [&] { hipLaunchKernelGGL(k_copy16, dim3(4096), dim3(256), 0, 0, keys, dst, n); });
    return 0;
}
