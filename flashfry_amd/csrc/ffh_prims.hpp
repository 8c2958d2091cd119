// ffh_prims.hpp -- device-wide primitives for gfx950 (wave64): exclusive scan and LSD radix sort of u64 keys.
// Hand-written; no rocPRIM/hipCUB.  Everything is launched on the caller's stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ffh {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---------------------------------------------------------------------------------------------------------
// exclusive scan:  out[i] = sum_{j<i} in[j],  out[n] = total   (out has n+1 entries; in-place allowed when
// TOut == TIn and out == in is NOT used -- callers pass distinct buffers).  Three-kernel reduce/scan/scatter,
// recursive over block sums.  Block = 256 threads x 16 items = 4096 items.
// ---------------------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T *lds /* >= 8 entries */, T &block_total) {
    // wave inclusive scan via DPP-free shuffles, then across the 4 waves through LDS
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    T wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        T s = lds[w];
        if ((uint32_t)w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    block_total = tot;
    return wave_off + incl - v;
}

// a thread's kScanItems consecutive inputs; full tiles are fetched with 16-byte loads
template <typename TIn, typename TOut>
__device__ __forceinline__ void scan_load_items(const TIn *__restrict__ in, uint64_t n, uint64_t base, TOut (&v)[kScanItems]) {
    if (base + kScanItems <= n) {
        constexpr int kPer = 16 / sizeof(TIn);
        struct alignas(16) Pack { TIn e[kPer]; };
        const Pack *src = reinterpret_cast<const Pack *>(in + base);
#pragma unroll
        for (int q = 0; q < kScanItems / kPer; ++q) {
            const Pack p = src[q];
#pragma unroll
            for (int e = 0; e < kPer; ++e) v[q * kPer + e] = (TOut)p.e[e];
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) v[k] = (base + k < n) ? (TOut)in[base + k] : (TOut)0;
    }
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(kScanThreads) void k_scan_reduce(const TIn *__restrict__ in, uint64_t n, TOut *__restrict__ bsum) {
    __shared__ TOut lds[8];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    TOut v[kScanItems];
    scan_load_items<TIn, TOut>(in, n, base, v);
    TOut s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) s += v[k];
    TOut tot;
    block_exclusive_scan<TOut>(s, lds, tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// SUMS: `boff` holds the raw block sums and every block adds up the ones before it itself (<= kScanInlineBlocks of them): the
// scans here are short and launch-bound, two launches instead of a recursion of five
constexpr uint64_t kScanInlineBlocks = 4096;
template <typename TIn, typename TOut, bool SUMS>
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const TIn *__restrict__ in, uint64_t n, const TOut *__restrict__ boff,
                                                              TOut *__restrict__ out) {
    __shared__ TOut lds[8];
    TOut before = 0;
    if (SUMS) {
        TOut part = 0;
        for (uint32_t j = threadIdx.x; j < blockIdx.x; j += kScanThreads) part += boff[j];
        TOut dummy;
        const TOut excl = block_exclusive_scan<TOut>(part, lds, dummy);
        (void)excl;
        before = dummy;  // block_exclusive_scan returns the block total through its last argument
    } else {
        before = boff[blockIdx.x];
    }
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    TOut v[kScanItems];
    scan_load_items<TIn, TOut>(in, n, base, v);
    TOut s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) s += v[k];
    TOut tot;
    TOut off = block_exclusive_scan<TOut>(s, lds, tot) + before;
    if (base + kScanItems <= n) {
        constexpr int kPer = 16 / sizeof(TOut);
        struct alignas(16) Pack { TOut e[kPer]; };
        Pack *dst = reinterpret_cast<Pack *>(out + base);
#pragma unroll
        for (int q = 0; q < kScanItems / kPer; ++q) {
            Pack p;
#pragma unroll
            for (int e = 0; e < kPer; ++e) { p.e[e] = off; off += v[q * kPer + e]; }
            dst[q] = p;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            if (base + k < n) out[base + k] = off;
            off += v[k];
        }
    }
    // the grand total goes to out[n]
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kScanThreads - 1) out[n] = off;
}

// out must hold n+1 entries.  scratch: scan_scratch_elems_safe(n) TOut elements.
template <typename TIn, typename TOut>
inline void exclusive_scan(const TIn *in, uint64_t n, TOut *out, TOut *scratch, hipStream_t st) {
    uint64_t nb = (n + kScanTile - 1) / kScanTile;
    if (nb == 0) nb = 1;
    const uint64_t nb_pad = (nb + 1 + 7) & ~7ull;  // keep every sub-buffer 16-byte aligned for the vector loads
    TOut *bsum = scratch;            // nb entries (+1 for the recursive total)
    TOut *next = scratch + nb_pad;   // scratch of the next level
    if (nb == 1) {  // one block: nothing before it
        hipLaunchKernelGGL((k_scan_apply<TIn, TOut, true>), dim3(1), dim3(kScanThreads), 0, st, in, n, (const TOut *)bsum, out);
        return;
    }
    hipLaunchKernelGGL((k_scan_reduce<TIn, TOut>), dim3((unsigned)nb), dim3(kScanThreads), 0, st, in, n, bsum);
    if (nb <= kScanInlineBlocks) {
        hipLaunchKernelGGL((k_scan_apply<TIn, TOut, true>), dim3((unsigned)nb), dim3(kScanThreads), 0, st, in, n, (const TOut *)bsum, out);
    } else {
        // scan the block sums: bsum -> boff (stored in `next` region's head), recursive
        TOut *boff = next;
        exclusive_scan<TOut, TOut>(bsum, nb, boff, next + nb_pad, st);
        hipLaunchKernelGGL((k_scan_apply<TIn, TOut, false>), dim3((unsigned)nb), dim3(kScanThreads), 0, st, in, n, (const TOut *)boff, out);
    }
}

// scratch needed when recursion allocates [bsum(nb+1)][boff(nb+1)][next level ...]
inline uint64_t scan_scratch_elems_safe(uint64_t n) {
    uint64_t tot = 16;
    while (true) {
        uint64_t nb = (n + kScanTile - 1) / kScanTile;
        if (nb == 0) nb = 1;
        tot += 2 * ((nb + 1 + 7) & ~7ull);
        if (nb <= 1) break;
        n = nb;
    }
    return tot;
}

// ---------------------------------------------------------------------------------------------------------
// LSD radix sort of u64 keys, 8 bits per pass, only over the caller-given bit ranges.
// A 256-thread block owns a contiguous chunk of 4096 keys.  Per pass: histogram -> scan of the digit-major
// (256 x nblocks) table -> scatter.  The scatter ranks the chunk stably (wave w owns rows w*16..w*16+15, ranked row
// by row with ballots), stages it digit-ordered in LDS and writes every digit's run contiguously, so the global
// stores are coalesced runs instead of 8-byte scatters.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSortThreads = 256;
constexpr int kSortRows = 16;                               // rows of 64 keys per wave
constexpr int kSortChunk = kSortThreads * kSortRows;        // 4096 keys per block

__global__ __launch_bounds__(kSortThreads) void k_sort_hist(const uint64_t *__restrict__ keys, uint64_t n, int shift,
                                                             uint32_t *__restrict__ table /* [256][nblocks] */, uint32_t nblocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kSortChunk;
#pragma unroll
    for (int r = 0; r < kSortRows; ++r) {
        const uint64_t i = base + (uint64_t)r * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    table[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// VALS: a u64 payload travels with every key (same stable permutation), staged through the same LDS buffer after the keys
template <bool VALS>
__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const uint64_t *__restrict__ keys, uint64_t *__restrict__ out, uint64_t n, int shift,
                                                                const uint32_t *__restrict__ offs /* scanned [256][nblocks] */, uint32_t nblocks,
                                                                const uint64_t *__restrict__ vals, uint64_t *__restrict__ vout) {
    __shared__ uint64_t staged[kSortChunk];      // the chunk, digit-ordered
    __shared__ uint32_t wave_cnt[4][256];        // per-wave digit counts -> per-wave start inside the digit's run
    __shared__ uint32_t dig_start[256];          // start of every digit's run inside the chunk
    __shared__ uint32_t dig_goff[256];           // global offset of every digit's run
    __shared__ uint32_t scan_lds[8];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * kSortChunk;
    const uint32_t here = (uint32_t)min((uint64_t)kSortChunk, n - base);
    // wave w owns keys [w*1024, w*1024 + 1024) of the chunk, 16 rows of 64
    uint64_t kreg[kSortRows];
    uint16_t preg[VALS ? kSortRows : 1];  // where each of this thread's keys went inside the staged chunk
#pragma unroll
    for (int r = 0; r < kSortRows; ++r) {
        const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
        kreg[r] = i < here ? keys[base + i] : 0;
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) wave_cnt[w][threadIdx.x] = 0;
    dig_goff[threadIdx.x] = offs[(uint64_t)threadIdx.x * nblocks + blockIdx.x];
    __syncthreads();
    // pass 1: per-wave digit counts
#pragma unroll
    for (int r = 0; r < kSortRows; ++r) {
        const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
        if (i < here) atomicAdd(&wave_cnt[wave][(uint32_t)(kreg[r] >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    // digit d (= threadIdx.x): run start inside the chunk, and each wave's start inside that run
    {
        const uint32_t c0 = wave_cnt[0][threadIdx.x], c1 = wave_cnt[1][threadIdx.x], c2 = wave_cnt[2][threadIdx.x], c3 = wave_cnt[3][threadIdx.x];
        uint32_t tot;
        const uint32_t start = block_exclusive_scan<uint32_t>(c0 + c1 + c2 + c3, scan_lds, tot);
        dig_start[threadIdx.x] = start;
        wave_cnt[0][threadIdx.x] = start;
        wave_cnt[1][threadIdx.x] = start + c0;
        wave_cnt[2][threadIdx.x] = start + c0 + c1;
        wave_cnt[3][threadIdx.x] = start + c0 + c1 + c2;
    }
    __syncthreads();
    // pass 2: stable rank inside the wave, row by row; wave_cnt[wave][d] is the wave's running cursor
#pragma unroll
    for (int r = 0; r < kSortRows; ++r) {
        const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
        const bool valid = i < here;
        const uint32_t d = (uint32_t)(kreg[r] >> shift) & 0xFF;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t bal = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? bal : ~bal;
        }
        const uint32_t rank = mbcnt(peers);
        uint32_t pos = 0;
        if (valid) pos = wave_cnt[wave][d] + rank;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == (uint32_t)__popcll(peers) - 1) wave_cnt[wave][d] = pos + 1;  // last peer advances the cursor
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (valid) staged[pos] = kreg[r];
        if (VALS) preg[VALS ? r : 0] = (uint16_t)pos;
    }
    __syncthreads();
    // pass 3: contiguous copy-out; element i of the digit-ordered chunk goes to the digit's global run
    if (!VALS) {
        for (uint32_t i = threadIdx.x; i < here; i += kSortThreads) {
            const uint64_t key = staged[i];
            const uint32_t d = (uint32_t)(key >> shift) & 0xFF;
            out[(uint64_t)dig_goff[d] + (i - dig_start[d])] = key;
        }
    } else {
        uint32_t dest[kSortRows];
#pragma unroll
        for (int k = 0; k < kSortRows; ++k) {
            const uint32_t i = threadIdx.x + k * kSortThreads;
            dest[k] = 0;
            if (i < here) {
                const uint64_t key = staged[i];
                const uint32_t d = (uint32_t)(key >> shift) & 0xFF;
                dest[k] = dig_goff[d] + (i - dig_start[d]);
                out[dest[k]] = key;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kSortRows; ++r) {
            const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
            if (i < here) staged[preg[VALS ? r : 0]] = vals[base + i];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSortRows; ++k) {
            const uint32_t i = threadIdx.x + k * kSortThreads;
            if (i < here) vout[dest[k]] = staged[i];
        }
    }
}

// up to kSmallSort keys: one block, bitonic network in LDS, in place (a handful of guides against a small database gives a few
// thousand hits; the multi-pass radix sort would spend 20 launches on them)
constexpr uint32_t kSmallSort = 4096;
__global__ __launch_bounds__(1024) void k_sort_small(uint64_t *__restrict__ keys, uint32_t n) {
    __shared__ uint64_t s[kSmallSort];
    for (uint32_t i = threadIdx.x; i < kSmallSort; i += blockDim.x) s[i] = i < n ? keys[i] : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= kSmallSort; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < kSmallSort; i += blockDim.x) {
                const uint32_t p = i ^ j;
                if (p > i) {
                    const uint64_t a = s[i], b = s[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[p] = a; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) keys[i] = s[i];
}

struct SortScratch {
    uint64_t *alt = nullptr;     // n keys
    uint64_t *val_alt = nullptr; // n payloads (radix_sort_pairs only)
    uint32_t *table = nullptr;   // 256*nblocks + 1
    uint32_t *offs = nullptr;    // 256*nblocks + 1
    uint32_t *scan_tmp = nullptr;
};

inline uint32_t sort_nblocks(uint64_t n) { return (uint32_t)((n + kSortChunk - 1) / kSortChunk); }

// sorts keys[0..n) ascending considering only bits [lo_a, hi_a) and [lo_b, hi_b) (lo_b >= hi_a); returns the
// pointer (keys or scratch.alt) that holds the sorted result.
inline uint64_t *radix_sort_u64(uint64_t *keys, uint64_t n, int lo_a, int hi_a, int lo_b, int hi_b, SortScratch &s, hipStream_t st) {
    if (n == 0) return keys;
    const uint32_t nb = sort_nblocks(n);
    uint64_t *src = keys, *dst = s.alt;
    int ranges[2][2] = {{lo_a, hi_a}, {lo_b, hi_b}};
    for (int rg = 0; rg < 2; ++rg)
        for (int shift = ranges[rg][0]; shift < ranges[rg][1]; shift += 8) {
            hipLaunchKernelGGL(k_sort_hist, dim3(nb), dim3(kSortThreads), 0, st, src, n, shift, s.table, nb);
            exclusive_scan<uint32_t, uint32_t>(s.table, (uint64_t)256 * nb, s.offs, s.scan_tmp, st);
            hipLaunchKernelGGL(k_sort_scatter<false>, dim3(nb), dim3(kSortThreads), 0, st, src, dst, n, shift, s.offs, nb, (const uint64_t *)nullptr,
                               (uint64_t *)nullptr);
            uint64_t *t = src; src = dst; dst = t;
        }
    return src;
}

// stable sort of (key, payload) pairs by bits [lo, hi) of the key; on return keys_out / vals_out name the buffers
// (the caller's or the scratch ones) that hold the result
inline void radix_sort_pairs(uint64_t *keys, uint64_t *vals, uint64_t n, int lo, int hi, SortScratch &s, hipStream_t st, uint64_t *&keys_out, uint64_t *&vals_out) {
    keys_out = keys; vals_out = vals;
    if (n == 0) return;
    const uint32_t nb = sort_nblocks(n);
    uint64_t *src = keys, *dst = s.alt, *vsrc = vals, *vdst = s.val_alt;
    for (int shift = lo; shift < hi; shift += 8) {
        hipLaunchKernelGGL(k_sort_hist, dim3(nb), dim3(kSortThreads), 0, st, src, n, shift, s.table, nb);
        exclusive_scan<uint32_t, uint32_t>(s.table, (uint64_t)256 * nb, s.offs, s.scan_tmp, st);
        hipLaunchKernelGGL(k_sort_scatter<true>, dim3(nb), dim3(kSortThreads), 0, st, src, dst, n, shift, s.offs, nb, (const uint64_t *)vsrc, vdst);
        uint64_t *t = src; src = dst; dst = t;
        t = vsrc; vsrc = vdst; vdst = t;
    }
    keys_out = src; vals_out = vsrc;
}

}  // namespace ffh
