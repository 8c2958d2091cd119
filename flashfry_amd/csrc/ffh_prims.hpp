// ffh_prims.hpp -- device-wide primitives for gfx950 (wave64): exclusive scan and LSD radix sort of u64 keys.
// Hand-written; no rocPRIM/hipCUB.  Everything is launched on the caller's stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

namespace ffh {

constexpr int kWave = 64;

// The library is written for gfx950 only: wave64, and 160 KB of LDS per workgroup -- k_msd_scatter stages 128 KB, k_binsort ~66 KB,
// k_slab_totals / k_slab_subhist 64 KB (static_asserts at the kernels).  Another --offload-arch (gfx942 and gfx90a stop at 64 KB) must
// not get as far as a launch failure:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "flashfry_hip is gfx950-only: build with --offload-arch=gfx950"
#endif
constexpr size_t kLdsPerBlock = 160 * 1024;

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
// Count tables of the radix passes: BLOCK-major, table[block][digit] -- a block leaves its counts as one contiguous run and reads its
// offsets the same way.  (Digit-major until round 5: 512 four-byte writes per block into lines shared with the neighbouring blocks cost
// a histogram pass over 4.7e7 keys 54 of its 128 us, tools/probes/bw_probe.hip; block-major costs 8.)  The offsets a pass needs are the
// exclusive scan of the counts in digit-major order: offs[b][d] = sum of all counts of digits < d + counts of digit d in blocks < b.
//   k_tab_colsum   per group of T blocks, the digit's count over the group  -> partial[digit][group]   (coalesced reads across digits)
//   exclusive_scan of partial (digits x groups, a few 1e5 entries)
//   k_tab_apply    per group: the running offset of every digit walked down the group's blocks      -> offs[block][digit]
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kTabThreads = 256, kTabGroupMin = 8;
inline uint32_t tab_group(uint32_t nblocks) { uint32_t t = kTabGroupMin; while ((nblocks + t - 1) / t > 512u) t *= 2u; return t; }
// elements a table buffer must hold for `nblocks` blocks of `digits` counts: the table, then partial, then its scan
inline size_t tab_elems(uint32_t nblocks, uint32_t digits) { return (size_t)digits * nblocks + 2 * ((size_t)digits * (nblocks / kTabGroupMin + 2) + 8) + 8; }

__global__ __launch_bounds__(kTabThreads) void k_tab_colsum(const uint32_t *__restrict__ table, uint32_t digits, uint32_t nblocks, uint32_t T, uint32_t ngroups,
                                                            uint32_t *__restrict__ partial) {
    const uint32_t g = blockIdx.x, b0 = g * T, b1 = min(nblocks, b0 + T);
    for (uint32_t d = threadIdx.x; d < digits; d += kTabThreads) {
        uint32_t sum = 0;
#pragma unroll 8
        for (uint32_t b = b0; b < b1; ++b) sum += table[(uint64_t)b * digits + d];
        partial[(uint64_t)d * ngroups + g] = sum;
    }
}
__global__ __launch_bounds__(kTabThreads) void k_tab_apply(const uint32_t *__restrict__ table, uint32_t digits, uint32_t nblocks, uint32_t T, uint32_t ngroups,
                                                           const uint32_t *__restrict__ pscan, uint32_t *__restrict__ offs) {
    const uint32_t g = blockIdx.x, b0 = g * T, b1 = min(nblocks, b0 + T);
    for (uint32_t d = threadIdx.x; d < digits; d += kTabThreads) {
        uint32_t run = pscan[(uint64_t)d * ngroups + g];
#pragma unroll 8
        for (uint32_t b = b0; b < b1; ++b) {
            const uint32_t v = table[(uint64_t)b * digits + d];
            offs[(uint64_t)b * digits + d] = run;
            run += v;
        }
    }
}
// table: tab_elems(nblocks, digits) elements, the counts in front; offs: digits x nblocks; scan_tmp: scan_scratch_elems_safe(digits x nblocks) or more
inline void tab_scan(uint32_t *table, uint32_t digits, uint32_t nblocks, uint32_t *offs, uint32_t *scan_tmp, hipStream_t st) {
    const uint32_t T = tab_group(nblocks), ng = (nblocks + T - 1) / T;
    uint32_t *partial = table + (((size_t)digits * nblocks + 7) & ~(size_t)7), *pscan = partial + (((size_t)digits * ng + 8) & ~(size_t)7);
    hipLaunchKernelGGL(k_tab_colsum, dim3(ng), dim3(kTabThreads), 0, st, (const uint32_t *)table, digits, nblocks, T, ng, partial);
    exclusive_scan<uint32_t, uint32_t>(partial, (uint64_t)digits * ng, pscan, scan_tmp, st);
    hipLaunchKernelGGL(k_tab_apply, dim3(ng), dim3(kTabThreads), 0, st, (const uint32_t *)table, digits, nblocks, T, ng, (const uint32_t *)pscan, offs);
}

// ---------------------------------------------------------------------------------------------------------
// LSD radix sort of u64 keys, BITS (8 or 9) bits per pass, only over the caller-given bit ranges.
// A 256-thread block owns a contiguous chunk of 4096 keys.  Per pass: histogram -> scan of the (nblocks x 2^BITS)
// count table (tab_scan) -> scatter.  The scatter ranks the chunk stably (wave w owns rows w*16..w*16+15, ranked row
// by row with ballots), stages it digit-ordered in LDS and writes every digit's run contiguously, so the global
// stores are coalesced runs instead of 8-byte scatters.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSortThreads = 256;
constexpr int kSortRows = 16;                               // rows of 64 keys per wave
constexpr int kSortChunk = kSortThreads * kSortRows;        // 4096 keys per block
constexpr int kSortMaxBits = 9;

template <int BITS>
__global__ __launch_bounds__(kSortThreads) void k_sort_hist(const uint64_t *__restrict__ keys, uint64_t n, int shift,
                                                             uint32_t *__restrict__ table /* [nblocks][2^BITS] */, uint32_t nblocks) {
    constexpr uint32_t DIG = 1u << BITS;
    __shared__ uint32_t h[DIG];
    for (uint32_t d = threadIdx.x; d < DIG; d += kSortThreads) h[d] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kSortChunk;
#pragma unroll
    for (int r = 0; r < kSortRows; ++r) {
        const uint64_t i = base + (uint64_t)r * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & (DIG - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < DIG; d += kSortThreads) table[(uint64_t)blockIdx.x * DIG + d] = h[d];
}

// VALS: a u64 payload travels with every key (same stable permutation), staged through the same LDS buffer after the keys
template <bool VALS, int BITS>
__global__ __launch_bounds__(kSortThreads) void k_sort_scatter(const uint64_t *__restrict__ keys, uint64_t *__restrict__ out, uint64_t n, int shift,
                                                                const uint32_t *__restrict__ offs /* scanned [nblocks][2^BITS] */, uint32_t nblocks,
                                                                const uint64_t *__restrict__ vals, uint64_t *__restrict__ vout) {
    constexpr uint32_t DIG = 1u << BITS, PER = DIG / kSortThreads;   // digits owned by a thread: 2t .. 2t + PER - 1
    static_assert(BITS >= 8 && BITS <= kSortMaxBits, "256 or 512 digits per pass");
    __shared__ uint64_t staged[kSortChunk];      // the chunk, digit-ordered
    __shared__ uint32_t wave_cnt[4][DIG];        // per-wave digit counts -> per-wave start inside the digit's run
    __shared__ uint32_t dig_start[DIG];          // start of every digit's run inside the chunk
    __shared__ uint32_t dig_goff[DIG];           // global offset of every digit's run
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
    for (uint32_t d = threadIdx.x; d < DIG; d += kSortThreads) {
#pragma unroll
        for (int w = 0; w < 4; ++w) wave_cnt[w][d] = 0;
        dig_goff[d] = offs[(uint64_t)blockIdx.x * DIG + d];
    }
    __syncthreads();
    // pass 1: per-wave digit counts
#pragma unroll
    for (int r = 0; r < kSortRows; ++r) {
        const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
        if (i < here) atomicAdd(&wave_cnt[wave][(uint32_t)(kreg[r] >> shift) & (DIG - 1u)], 1u);
    }
    __syncthreads();
    // digits PER t .. PER t + PER - 1 (t = threadIdx.x): run start inside the chunk, and each wave's start inside that run
    {
        uint32_t c[PER][4], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k)
#pragma unroll
            for (int w = 0; w < 4; ++w) { c[k][w] = wave_cnt[w][PER * threadIdx.x + k]; sum += c[k][w]; }
        uint32_t tot;
        uint32_t start = block_exclusive_scan<uint32_t>(sum, scan_lds, tot);
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t d = PER * threadIdx.x + k;
            dig_start[d] = start;
#pragma unroll
            for (int w = 0; w < 4; ++w) { wave_cnt[w][d] = start; start += c[k][w]; }
        }
    }
    __syncthreads();
    // pass 2: stable rank inside the wave, row by row; wave_cnt[wave][d] is the wave's running cursor
#pragma unroll
    for (int r = 0; r < kSortRows; ++r) {
        const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
        const bool valid = i < here;
        const uint32_t d = (uint32_t)(kreg[r] >> shift) & (DIG - 1u);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
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
            const uint32_t d = (uint32_t)(key >> shift) & (DIG - 1u);
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
                const uint32_t d = (uint32_t)(key >> shift) & (DIG - 1u);
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

// ---------------------------------------------------------------------------------------------------------
// Ordering the hits: keys = (guide << tbits) | database index.  A full LSD sort makes six device-wide passes over them.  The
// guide bits alone take two (9 + 8 bits for 100 000 guides); after them every guide's hits are contiguous, in arbitrary order, and a
// guide has ~116 of them -- few enough for ONE WAVE to order by ranking: every key counts the keys of its segment that are smaller
// (all pairs, the segment's low words broadcast from LDS), which is its position.  No exchange network, no dependent steps, one read
// and one write of the keys.  Segments beyond kSegWaveMax keys (guides inside repeat families) are listed and ordered by
// k_segsort_heavy, one block each, with an LSD sort of its own over the segment.
// ---------------------------------------------------------------------------------------------------------
// ---- a bitonic network over R registers per lane of one wave (64 R elements) ----
// element e of the sequence = register e >> 6, lane e & 63; runs ascend where (e & K) == 0 (K = the whole network: everywhere)
// N: elements the network orders (a power of two <= 64 R; the elements from N on are left alone -- all-ones sentinels in every use)
template <int R, int K, int J, int N = 64 * R>
__device__ __forceinline__ void bitonic_step(uint32_t (&x)[R], uint32_t lane) {
    if constexpr (J >= 64) {   // the partner is another register of the same lane
        constexpr int jj = J / 64;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if ((r & jj) == 0) {
                const bool up = K >= N ? true : ((r * 64) & K) == 0;
                const uint32_t lo = min(x[r], x[r | jj]), hi = max(x[r], x[r | jj]);
                x[r] = up ? lo : hi; x[r | jj] = up ? hi : lo;
            }
    } else {
        const bool lower = (lane & (uint32_t)J) == 0u;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool up = K >= N ? true : K < 64 ? (lane & (uint32_t)K) == 0u : ((r * 64) & K) == 0;
            const uint32_t y = (uint32_t)__shfl_xor((int)x[r], J, 64);
            x[r] = (lower == up) ? min(x[r], y) : max(x[r], y);
        }
    }
}
template <int R, int K, int J, int N = 64 * R>
__device__ __forceinline__ void bitonic_merge(uint32_t (&x)[R], uint32_t lane) {
    bitonic_step<R, K, J, N>(x, lane);
    if constexpr (J > 1) bitonic_merge<R, K, J / 2, N>(x, lane);
}
template <int R, int K = 2, int N = 64 * R>
__device__ __forceinline__ void bitonic_sort(uint32_t (&x)[R], uint32_t lane) {
    bitonic_merge<R, K, K / 2, N>(x, lane);
    if constexpr (K < N) bitonic_sort<R, K * 2, N>(x, lane);
}
constexpr uint32_t kSegWaveMax = 1024;
constexpr int kSegRows = kSegWaveMax / 64;

__global__ __launch_bounds__(256) void k_segsort(uint64_t *__restrict__ keys, const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_end,
                                                 uint32_t n_guides, int tbits, uint32_t *__restrict__ heavy_list, uint32_t *__restrict__ n_heavy) {
    __shared__ __attribute__((aligned(16))) uint32_t low[4][kSegWaveMax];
    // (wave, segment bounds and everything derived from them are wave-uniform: said explicitly, or the compiler predicates every loop
    // below per lane and turns v_readlane into a waterfall)
    const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), g = blockIdx.x * 4 + wave;
    if (g >= n_guides) return;
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_begin[g]), n = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_end[g]) - b;
    if (n <= 1u) return;
    if (n > kSegWaveMax) {
        if (lane == 0) heavy_list[atomicAdd(n_heavy, 1u)] = g;
        return;
    }
    const uint32_t mask = tbits >= 32 ? 0xFFFFFFFFu : (1u << tbits) - 1u;
    const uint32_t K = (n + 63u) >> 6;   // keys per lane (uniform)
    uint64_t k[kSegRows];
    uint32_t rank[kSegRows];
#pragma unroll
    for (int r = 0; r < kSegRows; ++r) {
        k[r] = ~0ull; rank[r] = 0;
        if ((uint32_t)r < K) {
            const uint32_t i = (uint32_t)r * 64u + lane;
            if (i < n) { k[r] = keys[b + i]; low[wave][i] = (uint32_t)k[r] & mask; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (K <= 4u) {
        // the usual segment (<= 256 keys): a bitonic network over the keys' low words in registers -- no memory round trip, ~230
        // instructions for 65 .. 128 keys (round 4 ranked every key against all others: four v_readlane + eight v_cmp per four keys)
        const uint64_t hi = k[0] & ~(uint64_t)mask;   // (lane 0 always holds a key of the segment: n >= 2)
        const uint64_t hi_u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hi >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hi);
        auto run = [&](auto rtag) {
            constexpr int R = decltype(rtag)::value;
            uint32_t x[R];
#pragma unroll
            for (int r = 0; r < R; ++r) x[r] = (uint32_t)r * 64u + lane < n ? ((uint32_t)k[r] & mask) : 0xFFFFFFFFu;
            bitonic_sort<R>(x, lane);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if ((uint32_t)r * 64u + lane < n) keys[b + (uint32_t)r * 64u + lane] = hi_u | x[r];
        };
        if (K <= 1u) run(std::integral_constant<int, 1>{});
        else if (K <= 2u) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 4>{});
        return;
    } else {
        // larger segments: the low words from LDS, four per (broadcast) read
        const uint32_t n4 = (n + 3u) & ~3u;
        for (uint32_t i = n + lane; i < n4; i += 64) low[wave][i] = 0xFFFFFFFFu;   // (tbits < 32: never below a real key's low word)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t j = 0; j < n4; j += 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&low[wave][j]);
#pragma unroll
            for (int r = 0; r < kSegRows; ++r)
                if ((uint32_t)r < K) {
                    const uint32_t a = (uint32_t)k[r] & mask;
                    rank[r] += (v.x < a ? 1u : 0u) + (v.y < a ? 1u : 0u) + (v.z < a ? 1u : 0u) + (v.w < a ? 1u : 0u);
                }
        }
    }
    // (the whole segment is in registers: writing it back in place is safe; the keys of a segment are distinct -- a (guide, target)
    // pair is found once -- so the ranks are a permutation)
#pragma unroll
    for (int r = 0; r < kSegRows; ++r)
        if ((uint32_t)r < K && (uint32_t)r * 64u + lane < n) keys[b + rank[r]] = k[r];
}

// LSD radix sort of keys[0 .. n) over their low `bits` bits by ONE block of 256 threads, 8 bits per pass, 4096 keys per step (the ranking
// of k_sort_scatter with the digits' running cursors kept in LDS), ping-pong between `keys` and `alt`; the result ends in `keys`.
struct BlockSortLds {
    uint32_t cursor[256];
    uint32_t wcnt[4][256];
    uint32_t scan_lds[8];
};
__device__ __forceinline__ void block_lsd_sort(uint64_t *__restrict__ keys, uint64_t *__restrict__ alt, uint32_t n, int bits, BlockSortLds &L) {
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint64_t *src = keys, *dst = alt;
    for (int shift = 0; shift < bits; shift += 8) {
        L.cursor[t] = 0;
        __syncthreads();
        for (uint32_t i = t; i < n; i += 256) atomicAdd(&L.cursor[(uint32_t)(src[i] >> shift) & 255u], 1u);
        __syncthreads();
        uint32_t tot;
        const uint32_t start = block_exclusive_scan<uint32_t>(L.cursor[t], L.scan_lds, tot);
        L.cursor[t] = start;
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n; c0 += kSortChunk) {
            const uint32_t here = min((uint32_t)kSortChunk, n - c0);
            uint64_t kreg[kSortRows];
#pragma unroll
            for (int r = 0; r < kSortRows; ++r) {
                const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
                kreg[r] = i < here ? src[c0 + i] : 0ull;
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) L.wcnt[w][t] = 0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kSortRows; ++r) {
                const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
                if (i < here) atomicAdd(&L.wcnt[wave][(uint32_t)(kreg[r] >> shift) & 255u], 1u);
            }
            __syncthreads();
            {   // digit t: where each wave's keys of this step go; the digit's cursor moves on
                const uint32_t c0w = L.wcnt[0][t], c1w = L.wcnt[1][t], c2w = L.wcnt[2][t], c3w = L.wcnt[3][t], base = L.cursor[t];
                L.wcnt[0][t] = base; L.wcnt[1][t] = base + c0w; L.wcnt[2][t] = base + c0w + c1w; L.wcnt[3][t] = base + c0w + c1w + c2w;
                L.cursor[t] = base + c0w + c1w + c2w + c3w;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kSortRows; ++r) {
                const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
                const bool valid = i < here;
                const uint32_t d = (uint32_t)(kreg[r] >> shift) & 255u;
                uint64_t peers = __ballot(valid);
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) {
                    const uint64_t bal = __ballot((d >> bb) & 1);
                    peers &= ((d >> bb) & 1) ? bal : ~bal;
                }
                const uint32_t rank = mbcnt(peers);
                uint32_t pos = 0;
                if (valid) pos = L.wcnt[wave][d] + rank;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (valid && rank == (uint32_t)__popcll(peers) - 1u) L.wcnt[wave][d] = pos + 1u;   // last peer advances the wave's cursor
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (valid) dst[pos] = kreg[r];
            }
            __syncthreads();
        }
        uint64_t *x = src; src = dst; dst = x;
    }
    if (src != keys)
        for (uint32_t i = t; i < n; i += 256) keys[i] = src[i];
    __syncthreads();
}

// one block per listed segment (a guide with more than kSegWaveMax raw hits): its keys ordered by the low `tbits` bits
__global__ __launch_bounds__(256) void k_segsort_heavy(uint64_t *__restrict__ keys, uint64_t *__restrict__ alt, const uint32_t *__restrict__ seg_begin,
                                                       const uint32_t *__restrict__ seg_end, const uint32_t *__restrict__ heavy_list,
                                                       const uint32_t *__restrict__ n_heavy, int tbits) {
    __shared__ BlockSortLds L;
    const uint32_t nh = *n_heavy;
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        const uint32_t g = heavy_list[h], b = seg_begin[g], n = seg_end[g] - b;
        block_lsd_sort(keys + b, alt + b, n, tbits, L);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Ordering the hits of a scan with a moderate number of them (round 5; the step the benchmark times): TWO passes over the keys instead of
// ~eight.
//   1. k_msd_hist / scan / k_msd_scatter: ONE most-significant-digit pass groups the keys by the top B <= 11 bits of the guide field into
//      2^B bins of a few thousand keys each.  Nothing downstream depends on the order inside a bin, so the scatter is not stable: a key's
//      place in its block's run of a digit is the value an LDS atomic returns (k_sort_scatter ranks every row with eight ballots).
//      16 384 keys per block: at 2048 digits a block still writes runs of ~8 keys.
//   2. k_binsort: one block per bin, the bin never leaves the CU.  The bin's keys are counted per guide in LDS (its guides are the
//      2^sub_bits consecutive ones the digit names), their database indices are put in guide order inside LDS, and every wave then
//      takes guides: it ranks the guide's indices against each other (all pairs: v_readlane for <= 128 keys, LDS broadcasts beyond),
//      which is the key's final place, and writes key and segment bounds.  This is k_segments + k_segsort + the second device-wide
//      pass in one launch, with the keys read once from memory.
// The all-ones padding keys of the compare waves' chunks are dropped by the first pass: behind it the array holds the scan's real hits only.
// A bin with more keys than the LDS arrays hold (a guide inside a repeat family) is listed and sorted by k_binsort_heavy through
// memory; a scan with more hits per bin than the arrays hold ON AVERAGE keeps the multi-pass path (hit_ordering_plan).
// ---------------------------------------------------------------------------------------------------------
constexpr int kMsdThreads = 1024;
constexpr int kMsdRows = 16;
constexpr int kMsdChunk = kMsdThreads * kMsdRows;          // 16 384 keys per block
constexpr int kMsdMaxBits = 11;
constexpr uint32_t kBinCap = 12288;                         // keys of a bin k_binsort holds in LDS (12 per thread)
constexpr int kBinRows = kBinCap / kMsdThreads;
constexpr int kBinMaxSubBits = 11;                          // guides per bin <= 2048

constexpr int kMsdRowsSmall = 4;
inline int msd_rows(uint64_t n) { return n >= (uint64_t)128 * kMsdChunk ? kMsdRows : kMsdRowsSmall; }   // (below 2.1e6 keys: <= 512 bins, a 4 096-key block still writes runs of ~8 keys)
inline uint32_t msd_nblocks(uint64_t n) { const uint64_t chunk = (uint64_t)msd_rows(n) * kMsdThreads; return (uint32_t)((n + chunk - 1) / chunk); }

// exclusive scan over the 1024 threads of a block
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t *lds /* >= 16 */, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t s = lds[w];
        if ((uint32_t)w < wave) off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return off + incl - v;
}

// (keys whose guide field is >= n_guides are the all-ones padding of the compare waves' last chunks: counted nowhere and dropped by the
// scatter, so the bins hold hits only and their total is the scan's number of real hits)
// ROWS: keys per thread = 16 (16 384 per block) for the large scans, 4 for the mid-size ones (1e6 keys in 71 blocks left most of the part idle)
template <int ROWS>
__global__ __launch_bounds__(kMsdThreads) void k_msd_hist(const uint64_t *__restrict__ keys, uint64_t n, int shift, uint32_t nbins, int tbits, uint32_t n_guides,
                                                           uint32_t *__restrict__ table /* [nblocks][nbins] */, uint32_t nblocks) {
    __shared__ uint32_t h[1 << kMsdMaxBits];
    for (uint32_t d = threadIdx.x; d < nbins; d += kMsdThreads) h[d] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * (ROWS * kMsdThreads);
    uint64_t kreg[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint64_t i = base + (uint64_t)r * kMsdThreads + threadIdx.x;
        kreg[r] = i < n ? keys[i] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
        if ((kreg[r] >> tbits) < n_guides) atomicAdd(&h[(uint32_t)(kreg[r] >> shift) & (nbins - 1u)], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nbins; d += kMsdThreads) table[(uint64_t)blockIdx.x * nbins + d] = h[d];
}

template <int ROWS>
__global__ __launch_bounds__(kMsdThreads) void k_msd_scatter(const uint64_t *__restrict__ keys, uint64_t *__restrict__ out, uint64_t n, int shift, uint32_t nbins, int tbits,
                                                              uint32_t n_guides, const uint32_t *__restrict__ offs /* scanned [nblocks][nbins] */, uint32_t nblocks) {
    constexpr int kMsdRows = ROWS;
    constexpr uint32_t kMsdChunk = (uint32_t)ROWS * kMsdThreads;
    __shared__ uint64_t staged[kMsdChunk];                  // the chunk, digit-ordered (128 KB at 16 rows)
    __shared__ uint32_t cnt[1 << kMsdMaxBits], dig_start[1 << kMsdMaxBits], dig_goff[1 << kMsdMaxBits];
    __shared__ uint32_t scan_lds[16];
    static_assert(sizeof staged + 3 * sizeof cnt + sizeof scan_lds <= kLdsPerBlock, "k_msd_scatter: the staged chunk must fit gfx950's 160 KB of LDS");
    const uint32_t t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * kMsdChunk;
    const uint32_t here = (uint32_t)min((uint64_t)kMsdChunk, n - base);
    uint64_t kreg[kMsdRows];
    uint16_t rank[kMsdRows];
#pragma unroll
    for (int r = 0; r < kMsdRows; ++r) {
        const uint32_t i = (uint32_t)r * kMsdThreads + t;
        kreg[r] = i < here ? keys[base + i] : ~0ull;
    }
    for (uint32_t d = t; d < nbins; d += kMsdThreads) { cnt[d] = 0; dig_goff[d] = offs[(uint64_t)blockIdx.x * nbins + d]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kMsdRows; ++r) {
        rank[r] = 0;
        if ((kreg[r] >> tbits) < n_guides) rank[r] = (uint16_t)atomicAdd(&cnt[(uint32_t)(kreg[r] >> shift) & (nbins - 1u)], 1u);
    }
    __syncthreads();
    uint32_t staged_n = 0;
    {   // digits 2t, 2t + 1: start of the digit's run inside the chunk
        const uint32_t c0 = 2u * t < nbins ? cnt[2u * t] : 0u, c1 = 2u * t + 1u < nbins ? cnt[2u * t + 1u] : 0u;
        const uint32_t s = block_scan_1024(c0 + c1, scan_lds, staged_n);
        if (2u * t < nbins) dig_start[2u * t] = s;
        if (2u * t + 1u < nbins) dig_start[2u * t + 1u] = s + c0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kMsdRows; ++r)
        if ((kreg[r] >> tbits) < n_guides) staged[dig_start[(uint32_t)(kreg[r] >> shift) & (nbins - 1u)] + rank[r]] = kreg[r];
    __syncthreads();
    for (uint32_t i = t; i < staged_n; i += kMsdThreads) {
        const uint64_t key = staged[i];
        const uint32_t d = (uint32_t)(key >> shift) & (nbins - 1u);
        out[(uint64_t)dig_goff[d] + (i - dig_start[d])] = key;
    }
}

// One wave orders the c distinct 32-bit database indices idx[0 .. c) of a guide and writes the keys hi | index in that order.
// c <= 256: a bitonic network over one, two or four registers per lane -- element e of the sequence lives in lane e & 63 of register
// e >> 6, absent elements are all-ones and end up behind the others; the partner of a compare-exchange comes through the LDS crossbar
// (ds_bpermute, no memory access) or is the lane's other register: ~230 instructions for a guide of 65 .. 128 keys.  (Round 4 ranked every key against all others -- c / 4 steps of four v_readlane and
// eight half-rate v_cmp: ~700 instructions for the usual 116 keys, 0.15 ms of the hg38-scale step on the vector pipes alone.)
// Beyond 256 keys: ranking against the whole segment out of LDS, four keys per broadcast step, the lane's own keys in chunks of
// kBinRankRows x 64.
constexpr int kBinRankRows = 8;
template <int R>
__device__ __forceinline__ void sort_segment_regs(const uint32_t *__restrict__ idx, uint32_t c, uint64_t hi, uint64_t *__restrict__ keys, uint32_t lane) {
    uint32_t x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = (uint32_t)r * 64u + lane < c ? idx[(uint32_t)r * 64u + lane] : 0xFFFFFFFFu;
    if constexpr (R == 1) {   // (a guide with a handful of hits -- a chr22-scale call has 1.7 per guide: a network of its size, not of 64)
        if (c <= 2u) bitonic_sort<1, 2, 2>(x, lane);
        else if (c <= 4u) bitonic_sort<1, 2, 4>(x, lane);
        else if (c <= 8u) bitonic_sort<1, 2, 8>(x, lane);
        else if (c <= 16u) bitonic_sort<1, 2, 16>(x, lane);
        else if (c <= 32u) bitonic_sort<1, 2, 32>(x, lane);
        else bitonic_sort<1>(x, lane);
    } else bitonic_sort<R>(x, lane);
#pragma unroll
    for (int r = 0; r < R; ++r)
        if ((uint32_t)r * 64u + lane < c) keys[(uint32_t)r * 64u + lane] = hi | x[r];
}
__device__ __forceinline__ void rank_segment(const uint32_t *__restrict__ idx, uint32_t c, uint64_t hi, uint64_t *__restrict__ keys, uint32_t lane) {
    if (c == 1u) { if (lane == 0) keys[0] = hi | idx[0]; return; }
    if (c <= 64u) { sort_segment_regs<1>(idx, c, hi, keys, lane); return; }
    if (c <= 128u) { sort_segment_regs<2>(idx, c, hi, keys, lane); return; }
    if (c <= 256u) { sort_segment_regs<4>(idx, c, hi, keys, lane); return; }
}
// keys [c0, c0 + 64 ROWS) of a segment of c > 256 indices ranked against the whole segment out of LDS, four per broadcast step (the
// rare large guide of a dense scan, by the wave that owns it; k_binsort<true> has the form for scans MADE of large guides)
template <int ROWS>
__device__ __forceinline__ void rank_unit(const uint32_t *__restrict__ idx, uint32_t c, uint32_t c0, uint64_t hi, uint64_t *__restrict__ keys, uint32_t lane) {
    uint32_t a[ROWS], rank[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint32_t i = c0 + (uint32_t)r * 64u + lane;
        a[r] = i < c ? idx[i] : 0xFFFFFFFFu;
        rank[r] = 0;
    }
    for (uint32_t j = 0; j < c; j += 4) {   // (uniform addresses: LDS broadcasts)
        const uint32_t v0 = idx[j], v1 = j + 1u < c ? idx[j + 1u] : 0xFFFFFFFFu, v2 = j + 2u < c ? idx[j + 2u] : 0xFFFFFFFFu, v3 = j + 3u < c ? idx[j + 3u] : 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) rank[r] += (v0 < a[r] ? 1u : 0u) + (v1 < a[r] ? 1u : 0u) + (v2 < a[r] ? 1u : 0u) + (v3 < a[r] ? 1u : 0u);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
        if (c0 + (uint32_t)r * 64u + lane < c) keys[rank[r]] = hi | a[r];
}

// bin d = keys [offs[d], offs[d + 1]) of `keys` (row 0 of the table k_msd_scatter used: block 0's offset of a digit is where the
// digit's run begins); in place.
// COOP: guides with more than 256 hits are set aside and ordered by all 16 waves together after the rest -- the form for scans whose
// guides are few and large (ten guides of 1 140 hits each: one wave per guide left fifteen idle for 0.25 ms); without it each wave ranks
// its own guides whatever their size and leaves as soon as it is done (the hg38-scale scan: 116 hits a guide; the barrier and the second
// phase cost 20 of 122 us there, profiles/r05/ab_log.txt item 9)
template <bool COOP>
__global__ __launch_bounds__(kMsdThreads) void k_binsort(uint64_t *__restrict__ keys, const uint32_t *__restrict__ offs, uint32_t nblocks, uint32_t nbins, uint64_t n_total,
                                                          int tbits, int sub_bits, uint32_t n_guides, uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end,
                                                          uint32_t *__restrict__ heavy_list, uint32_t *__restrict__ n_heavy) {
    __shared__ uint32_t idx[kBinCap];
    __shared__ uint32_t cnt[(1 << kBinMaxSubBits) + 2], start[(1 << kBinMaxSubBits) + 2];
    __shared__ uint32_t scan_lds[16];
    __shared__ uint32_t big[COOP ? kBinCap / 256 + 2 : 1], n_big;   // the bin's guides with more than 256 hits: ranked by all waves together (below)
    static_assert(sizeof idx + sizeof cnt + sizeof start + sizeof scan_lds + sizeof big + 4 <= kLdsPerBlock, "k_binsort: a bin's keys must fit gfx950's 160 KB of LDS (a smaller kBinCap elsewhere)");
    // (the cross-chunk ranking below bisects with a strict `<`: it relies on the (guide, database index) records of a scan being unique, which the
    // compare launch guarantees -- a pair within the prefix radius is the prefix image's to report and nobody else's, ffh_compare.hpp)
    const uint32_t t = threadIdx.x, lane = t & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6)), bin = blockIdx.x;
    if (COOP && t == 0) n_big = 0;
    // offs == nullptr: ONE bin = all n_total records as the compare launch left them, chunk padding included (a small scan: this launch
    // is the whole ordering); the padding is dropped as the records are read
    const uint32_t b0 = offs ? offs[bin] : 0u, b1 = !offs || bin + 1u >= nbins ? (uint32_t)n_total : offs[bin + 1u], n = b1 - b0;
    if (n == 0u) return;
    if (n > kBinCap) {
        if (t == 0) heavy_list[atomicAdd(n_heavy, 1u)] = bin;
        return;
    }
    const uint32_t nsub = 1u << sub_bits, mask = tbits >= 32 ? 0xFFFFFFFFu : (1u << tbits) - 1u;
    for (uint32_t s = t; s <= nsub; s += kMsdThreads) cnt[s] = 0;
    __syncthreads();
    uint64_t kreg[kBinRows];
    uint16_t rk[kBinRows];
    auto sub_of = [&](uint64_t key) { const uint32_t g = (uint32_t)(key >> tbits); return g >= n_guides ? nsub : g & (nsub - 1u); };   // (nsub: chunk padding, dropped)
#pragma unroll
    for (int r = 0; r < kBinRows; ++r) {
        const uint32_t i = (uint32_t)r * kMsdThreads + t;
        kreg[r] = 0; rk[r] = 0;
        if (i < n) { kreg[r] = keys[b0 + i]; rk[r] = (uint16_t)atomicAdd(&cnt[sub_of(kreg[r])], 1u); }
    }
    __syncthreads();
    {   // exclusive scan of the nsub + 1 counters (<= 2049: three per thread at most)
        uint32_t c[3], sum = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const uint32_t s = 3u * t + (uint32_t)k; c[k] = s <= nsub ? cnt[s] : 0u; sum += c[k]; }
        uint32_t tot;
        uint32_t off = block_scan_1024(sum, scan_lds, tot);
#pragma unroll
        for (int k = 0; k < 3; ++k) { const uint32_t s = 3u * t + (uint32_t)k; if (s <= nsub) start[s] = off; off += c[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kBinRows; ++r) {
        const uint32_t i = (uint32_t)r * kMsdThreads + t;
        if (i < n) idx[start[sub_of(kreg[r])] + rk[r]] = (uint32_t)kreg[r] & mask;   // (padding: behind the last guide's indices, never read)
    }
    __syncthreads();
    // every wave takes guides of the bin; all of the bin's keys are in registers / LDS by now, so writing in place is safe
    for (uint32_t s = wave; s < nsub; s += kMsdThreads / 64) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[s]);
        if (c == 0u) continue;
        const uint32_t st = (uint32_t)__builtin_amdgcn_readfirstlane((int)start[s]), guide = (bin << sub_bits) | s;
        if (lane == 0) { seg_begin[guide] = b0 + st; seg_end[guide] = b0 + st + c; }
        if (c <= 256u) rank_segment(idx + st, c, (uint64_t)guide << tbits, keys + b0 + st, lane);
        else if (COOP) { if (lane == 0) big[atomicAdd(&n_big, 1u)] = s; }
        else for (uint32_t c0 = 0; c0 < c; c0 += 512u) rank_unit<8>(idx + st, c, c0, (uint64_t)guide << tbits, keys + b0 + st, lane);
    }
    if (!COOP) return;
    __syncthreads();
    // the large guides, in chunks of 256 indices dealt round-robin to the 16 waves: every chunk ordered in registers and put back (1),
    // then every index ranked = its place in its own chunk + the number of smaller indices in each other chunk of the guide, found by
    // bisection (2): 9 LDS reads per other chunk and index -- ranking against every index of the guide, four per broadcast step, cost
    // 285 steps of 8 reads for a guide of 1 140 hits, 0.28 ms for ten of them in one block
    const uint32_t nb = n_big;
    auto for_chunks = [&](auto &&fn) {
        uint32_t i = 0, first = 0;   // chunks of big[0 .. i) = first
        for (uint32_t u = wave; i < nb; u += kMsdThreads / 64) {
            uint32_t c = 0, s = 0;
            for (; i < nb; ++i) {
                s = big[i]; c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[s]);
                const uint32_t units = (c + 255u) / 256u;
                if (u < first + units) break;
                first += units;
            }
            if (i >= nb) break;
            fn(s, c, (uint32_t)__builtin_amdgcn_readfirstlane((int)start[s]), (u - first) * 256u);
        }
    };
    for_chunks([&](uint32_t, uint32_t c, uint32_t st, uint32_t c0) {
        uint32_t x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = c0 + (uint32_t)r * 64u + lane < c ? idx[st + c0 + (uint32_t)r * 64u + lane] : 0xFFFFFFFFu;
        bitonic_sort<4>(x, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c0 + (uint32_t)r * 64u + lane < c) idx[st + c0 + (uint32_t)r * 64u + lane] = x[r];
    });
    __syncthreads();
    for_chunks([&](uint32_t s, uint32_t c, uint32_t st, uint32_t c0) {
        uint32_t x[4], rank[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t e = c0 + (uint32_t)r * 64u + lane;
            x[r] = e < c ? idx[st + e] : 0xFFFFFFFFu;
            rank[r] = (uint32_t)r * 64u + lane;
        }
        for (uint32_t q0 = 0; q0 < c; q0 += 256u) {
            if (q0 == c0) continue;
            const uint32_t *__restrict__ other = idx + st + q0;
            const uint32_t lq = min(256u, c - q0);
            uint32_t pos[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (uint32_t step = 256u; step; step >>= 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t p = pos[r] + step;
                    if (p <= lq && other[p - 1u] < x[r]) pos[r] = p;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) rank[r] += pos[r];
        }
        const uint64_t hi = (uint64_t)((bin << sub_bits) | s) << tbits;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c0 + (uint32_t)r * 64u + lane < c) keys[b0 + st + rank[r]] = hi | x[r];
    });
}

// the bins k_binsort listed: sorted through memory by their low tbits + sub_bits bits, then the guides' segments found in the sorted run
__global__ __launch_bounds__(256) void k_binsort_heavy(uint64_t *__restrict__ keys, uint64_t *__restrict__ alt, const uint32_t *__restrict__ offs, uint32_t nblocks, uint32_t nbins,
                                                       uint64_t n_total, const uint32_t *__restrict__ heavy_list, const uint32_t *__restrict__ n_heavy, int tbits, int sub_bits,
                                                       uint32_t n_guides, uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end) {
    __shared__ BlockSortLds L;
    const uint32_t nh = *n_heavy;
    for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
        const uint32_t bin = heavy_list[h];
        const uint32_t b0 = offs[bin], b1 = bin + 1u < nbins ? offs[bin + 1u] : (uint32_t)n_total, n = b1 - b0;
        uint64_t *k = keys + b0;
        block_lsd_sort(k, alt + b0, n, tbits + sub_bits, L);
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint64_t g = k[i] >> tbits;
            if (g >= n_guides) continue;
            if (i == 0 || (k[i - 1] >> tbits) != g) seg_begin[g] = b0 + i;
            if (i == n - 1 || (k[i + 1] >> tbits) != g) seg_end[g] = b0 + i + 1u;
        }
        __syncthreads();
    }
}

struct SortScratch {
    uint64_t *alt = nullptr;     // n keys
    uint64_t *val_alt = nullptr; // n payloads (radix_sort_pairs only)
    uint32_t *table = nullptr;   // 512*nblocks + 1
    uint32_t *offs = nullptr;    // 512*nblocks + 1
    uint32_t *scan_tmp = nullptr;
    unsigned long long *status = nullptr;   // 512 x nblocks words: the one-sweep passes' look-back words (radix_sort_u64; null: the three-launch passes)
};
constexpr uint32_t kSortTableDigits = 1u << kSortMaxBits;   // table / offs hold this many digits x nblocks (+ 1)

inline uint32_t sort_nblocks(uint64_t n) { return (uint32_t)((n + kSortChunk - 1) / kSortChunk); }

// one stable pass over `bits` (8 or 9) bits from `shift` on: src -> dst
template <bool VALS>
inline void radix_pass(const uint64_t *src, uint64_t *dst, const uint64_t *vsrc, uint64_t *vdst, uint64_t n, int shift, int bits, SortScratch &s, hipStream_t st) {
    const uint32_t nb = sort_nblocks(n);
    if (bits > 8) {
        hipLaunchKernelGGL(k_sort_hist<9>, dim3(nb), dim3(kSortThreads), 0, st, src, n, shift, s.table, nb);
        tab_scan(s.table, 512u, nb, s.offs, s.scan_tmp, st);
        hipLaunchKernelGGL((k_sort_scatter<VALS, 9>), dim3(nb), dim3(kSortThreads), 0, st, src, dst, n, shift, s.offs, nb, vsrc, vdst);
    } else {
        hipLaunchKernelGGL(k_sort_hist<8>, dim3(nb), dim3(kSortThreads), 0, st, src, n, shift, s.table, nb);
        tab_scan(s.table, 256u, nb, s.offs, s.scan_tmp, st);
        hipLaunchKernelGGL((k_sort_scatter<VALS, 8>), dim3(nb), dim3(kSortThreads), 0, st, src, dst, n, shift, s.offs, nb, vsrc, vdst);
    }
}

// passes needed for a range of `len` bits with digits of at most 9 bits, and the width of pass k (the widths differ by at most one)
inline int radix_passes(int len) { return len <= 0 ? 0 : (len + kSortMaxBits - 1) / kSortMaxBits; }
inline int radix_width(int len, int k) { const int p = radix_passes(len); return len / p + (k < len % p ? 1 : 0); }

// ---------------------------------------------------------------------------------------------------------
// (round 5) The LSD sort without its histogram passes ("one sweep").  A pass of the sort above reads the keys twice -- k_sort_hist for the
// per-block digit counts, a device-wide scan of the 512 x nblocks table, then k_sort_scatter -- although the keys' digits of ALL passes
// are known before the first one.  Here ONE launch counts every pass's digits over the whole array (k_os_hist -> k_os_bases: where each
// digit's run begins, per pass), and a pass is a single launch: a block counts its own 4096 keys per digit, publishes the counts, and
// learns how many keys of each digit the blocks before it hold by looking BACK through their published words (decoupled look-back:
// a word is either a block's own count or already the inclusive sum up to that block, so the walk is short).  Blocks are dispatched in
// index order, so a block only ever waits for blocks that are resident or done.  Same stable ranking inside the block as
// k_sort_scatter; 1 + P launches and P + 1 reads of the keys instead of 4 P launches and 2 P reads.
// Status word: [63:48] pass tag (a word of another pass, or of the memset, is "not there yet"), [33:32] 1 = count, 2 = inclusive sum,
// [31:0] the value.
// ---------------------------------------------------------------------------------------------------------
constexpr int kOsMaxPasses = 8;
constexpr int kOsHistBlocks = 256;
constexpr size_t kOsPartialWords = (size_t)kOsHistBlocks * kOsMaxPasses * 512;   // what SortScratch::table has to hold for k_os_hist (offs: kOsMaxPasses x 512)
struct OsPlan { int n_passes; int shift[kOsMaxPasses]; int bits[kOsMaxPasses]; };

__global__ __launch_bounds__(1024) void k_os_hist(const uint64_t *__restrict__ keys, uint64_t n, OsPlan plan, uint32_t *__restrict__ partial /* [blocks][passes][512] */) {
    __shared__ uint32_t h[kOsMaxPasses][512];
    for (uint32_t d = threadIdx.x; d < (uint32_t)plan.n_passes * 512u; d += 1024) (&h[0][0])[d] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 1024) {
        const uint64_t k = keys[i];
        for (int p = 0; p < plan.n_passes; ++p) atomicAdd(&h[p][(uint32_t)(k >> plan.shift[p]) & ((1u << plan.bits[p]) - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < (uint32_t)plan.n_passes * 512u; d += 1024) partial[(size_t)blockIdx.x * plan.n_passes * 512u + d] = (&h[0][0])[d];
}
// one block per pass: base[pass][d] = keys with a smaller digit in that pass
__global__ __launch_bounds__(512) void k_os_bases(const uint32_t *__restrict__ partial, uint32_t n_blocks, int n_passes, uint32_t *__restrict__ base /* [passes][512] */) {
    __shared__ uint32_t wsum[8];
    const uint32_t p = blockIdx.x, d = threadIdx.x, lane = d & 63, wave = d >> 6;
    uint32_t c = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) c += partial[((size_t)b * n_passes + p) * 512u + d];
    uint32_t incl = c;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        const uint32_t o = __shfl_up(incl, k, 64);
        if (lane >= (uint32_t)k) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < wave; ++w) off += wsum[w];
    base[p * 512u + d] = off + incl - c;
}

template <int BITS>
__global__ __launch_bounds__(kSortThreads) void k_os_scatter(const uint64_t *__restrict__ keys, uint64_t *__restrict__ out, uint64_t n, int shift, const uint32_t *__restrict__ base /* [512] of this pass */,
                                                              unsigned long long *__restrict__ status /* [nblocks][512] */, uint32_t tag) {
    constexpr uint32_t DIG = 1u << BITS, PER = DIG / kSortThreads;
    static_assert(BITS >= 8 && BITS <= kSortMaxBits, "256 or 512 digits per pass");
    __shared__ uint64_t staged[kSortChunk];
    __shared__ uint32_t wave_cnt[4][DIG];
    __shared__ uint32_t dig_start[DIG];
    __shared__ uint32_t dig_goff[DIG];
    __shared__ uint32_t scan_lds[8];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t chunk0 = (uint64_t)blockIdx.x * kSortChunk;
    const uint32_t here = (uint32_t)min((uint64_t)kSortChunk, n - chunk0);
    uint64_t kreg[kSortRows];
#pragma unroll
    for (int r = 0; r < kSortRows; ++r) {
        const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
        kreg[r] = i < here ? keys[chunk0 + i] : 0;
    }
    for (uint32_t d = threadIdx.x; d < DIG; d += kSortThreads) {
#pragma unroll
        for (int w = 0; w < 4; ++w) wave_cnt[w][d] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortRows; ++r) {
        const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
        if (i < here) atomicAdd(&wave_cnt[wave][(uint32_t)(kreg[r] >> shift) & (DIG - 1u)], 1u);
    }
    __syncthreads();
    {
        uint32_t c[PER][4], tot_d[PER], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            tot_d[k] = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { c[k][w] = wave_cnt[w][PER * threadIdx.x + k]; tot_d[k] += c[k][w]; }
            sum += tot_d[k];
        }
        // publish this block's digit counts (block 0: they already are inclusive sums), then look back
        const unsigned long long tagw = (unsigned long long)tag << 48;
        unsigned long long *mine = status + (size_t)blockIdx.x * DIG;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k)
            __hip_atomic_store(mine + PER * threadIdx.x + k, tagw | ((blockIdx.x == 0 ? 2ull : 1ull) << 32) | tot_d[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t tot;
        uint32_t start = block_exclusive_scan<uint32_t>(sum, scan_lds, tot);
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t d = PER * threadIdx.x + k;
            uint32_t before = 0;   // keys of digit d in the blocks before this one
            // (four predecessors' words requested together: the walk is a chain of memory round trips)
            for (int64_t b = (int64_t)blockIdx.x - 1; b >= 0;) {
                unsigned long long v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    v[k] = b - k >= 0 ? __hip_atomic_load(status + (size_t)(b - k) * DIG + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tagw | (2ull << 32));
                bool done = false;
                int used = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (done || (v[k] >> 48) != tag) break;
                    before += (uint32_t)v[k];
                    ++used;
                    if (((v[k] >> 32) & 3ull) == 2ull) done = true;
                }
                if (done) break;
                b -= used;
            }
            if (blockIdx.x) __hip_atomic_store(mine + d, tagw | (2ull << 32) | (unsigned long long)(before + tot_d[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dig_goff[d] = base[d] + before;
            dig_start[d] = start;
#pragma unroll
            for (int w = 0; w < 4; ++w) { wave_cnt[w][d] = start; start += c[k][w]; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortRows; ++r) {
        const uint32_t i = wave * (kSortRows * 64) + r * 64 + lane;
        const bool valid = i < here;
        const uint32_t d = (uint32_t)(kreg[r] >> shift) & (DIG - 1u);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const uint64_t bal = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? bal : ~bal;
        }
        const uint32_t rank = mbcnt(peers);
        uint32_t pos = 0;
        if (valid) pos = wave_cnt[wave][d] + rank;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == (uint32_t)__popcll(peers) - 1) wave_cnt[wave][d] = pos + 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (valid) staged[pos] = kreg[r];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < here; i += kSortThreads) {
        const uint64_t key = staged[i];
        const uint32_t d = (uint32_t)(key >> shift) & (DIG - 1u);
        out[(uint64_t)dig_goff[d] + (i - dig_start[d])] = key;
    }
}

// sorts keys[0..n) ascending considering only bits [lo_a, hi_a) and [lo_b, hi_b) (lo_b >= hi_a); returns the
// pointer (keys or scratch.alt) that holds the sorted result.
inline uint64_t *radix_sort_u64(uint64_t *keys, uint64_t n, int lo_a, int hi_a, int lo_b, int hi_b, SortScratch &s, hipStream_t st) {
    if (n == 0) return keys;
    uint64_t *src = keys, *dst = s.alt;
    int ranges[2][2] = {{lo_a, hi_a}, {lo_b, hi_b}};
    if (s.status && n < (1ull << 32)) {   // one sweep: all passes' digit counts in one launch, one launch per pass
        OsPlan plan{};
        for (int rg = 0; rg < 2; ++rg) {
            const int len = ranges[rg][1] - ranges[rg][0];
            int shift = ranges[rg][0];
            for (int k = 0; k < radix_passes(len); ++k) {
                const int w = radix_width(len, k);
                plan.shift[plan.n_passes] = shift; plan.bits[plan.n_passes] = std::max(w, 8); ++plan.n_passes;
                shift += w;
            }
        }
        if (plan.n_passes <= kOsMaxPasses) {
            const uint32_t nb = sort_nblocks(n), hb = (uint32_t)std::min<uint64_t>(kOsHistBlocks, (n + 1023) / 1024);
            uint32_t *partial = s.table, *base = s.offs;   // (both hold >= 512 x nblocks words)
            (void)hipMemsetAsync(s.status, 0, (size_t)nb * 512 * sizeof(unsigned long long), st);
            hipLaunchKernelGGL(k_os_hist, dim3(hb), dim3(1024), 0, st, (const uint64_t *)keys, n, plan, partial);
            hipLaunchKernelGGL(k_os_bases, dim3(plan.n_passes), dim3(512), 0, st, (const uint32_t *)partial, hb, plan.n_passes, base);
            for (int p = 0; p < plan.n_passes; ++p) {
                if (plan.bits[p] > 8) hipLaunchKernelGGL(k_os_scatter<9>, dim3(nb), dim3(kSortThreads), 0, st, (const uint64_t *)src, dst, n, plan.shift[p], (const uint32_t *)base + p * 512, s.status, (uint32_t)p + 1u);
                else hipLaunchKernelGGL(k_os_scatter<8>, dim3(nb), dim3(kSortThreads), 0, st, (const uint64_t *)src, dst, n, plan.shift[p], (const uint32_t *)base + p * 512, s.status, (uint32_t)p + 1u);
                uint64_t *t = src; src = dst; dst = t;
            }
            return src;
        }
    }
    for (int rg = 0; rg < 2; ++rg) {
        const int len = ranges[rg][1] - ranges[rg][0];
        int shift = ranges[rg][0];
        for (int k = 0; k < radix_passes(len); ++k) {
            const int w = radix_width(len, k);
            radix_pass<false>(src, dst, nullptr, nullptr, n, shift, std::max(w, 8), s, st);   // (a digit narrower than 8 bits reads bits above the range: zeros or already-sorted bits)
            shift += w;
            uint64_t *t = src; src = dst; dst = t;
        }
    }
    return src;
}

// stable sort of (key, payload) pairs by bits [lo, hi) of the key; on return keys_out / vals_out name the buffers
// (the caller's or the scratch ones) that hold the result
inline void radix_sort_pairs(uint64_t *keys, uint64_t *vals, uint64_t n, int lo, int hi, SortScratch &s, hipStream_t st, uint64_t *&keys_out, uint64_t *&vals_out) {
    keys_out = keys; vals_out = vals;
    if (n == 0) return;
    uint64_t *src = keys, *dst = s.alt, *vsrc = vals, *vdst = s.val_alt;
    const int len = hi - lo;
    int shift = lo;
    for (int k = 0; k < radix_passes(len); ++k) {
        const int w = radix_width(len, k);
        radix_pass<true>(src, dst, vsrc, vdst, n, shift, std::max(w, 8), s, st);
        shift += w;
        uint64_t *t = src; src = dst; dst = t;
        t = vsrc; vsrc = vdst; vdst = t;
    }
    keys_out = src; vals_out = vsrc;
}

}  // namespace ffh
