// ffh_kernels.hpp -- the discover scan on gfx950: bucketed scan images, candidate lists, the compare kernel and
// the cut-off / scoring epilogue.  Integer XOR+popcount work on wave64; no MFMA (this is not a contraction).
//
// Encoding used on the device ("planar"): a target/guide long (bitcoding/BitEncoding.scala:46-67: 2 bits per base,
// interleaved) is split into its high-bit plane H and low-bit plane L restricted to the compared bases
// (ParameterPack.comparisonBitEncoding, standards/StandardScanParameters.scala:99,121,143,165,187,205).  With
// base i of the Lc compared bases at plane bit (Lc-1-i):
//        mismatches(g, t) = popcount( (Hg ^ Ht) | (Lg ^ Lt) )
// which is bit-for-bit BitEncoding.mismatches (:127-132) -- two XORs, one OR, one v_bcnt instead of the
// fold-onto-the-high-bit sequence.  A planar key is stored as u64 = (H << 32) | L.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ffh_prims.hpp"

namespace ffh {

struct Geometry {      // how the compared bases sit inside the 48-bit string field
    int c0;            // plane bit of the LAST compared base (Cas9: 3, Cpf1: 0)
    int lc;            // number of compared bases (20, or 19 for the 19-mer enzymes)
    int scan_len;      // bases per site (23, 22 or 24)
    int cas9_23;       // CFD / Hsu2013 defined
};

__host__ __device__ __forceinline__ uint32_t compress_even_bits(uint64_t x) {  // bits 0,2,4,..,46 -> 0..23
    x &= 0x5555555555555555ULL;
    x = (x | (x >> 1)) & 0x3333333333333333ULL;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
    return (uint32_t)x;
}

__host__ __device__ __forceinline__ uint64_t planar_key(uint64_t enc, int c0, int lc) {
    const uint32_t m = (1u << lc) - 1u;
    const uint32_t lo = (compress_even_bits(enc) >> c0) & m;
    const uint32_t hi = (compress_even_bits(enc >> 1) >> c0) & m;
    return ((uint64_t)hi << 32) | lo;
}

// bucket id over the first `a` compared bases / over the last `s` compared bases
__host__ __device__ __forceinline__ uint32_t prefix_bucket(uint64_t pk, int lc, int a) {
    if (a == 0) return 0;
    const uint32_t hi = (uint32_t)(pk >> 32) >> (lc - a), lo = (uint32_t)pk >> (lc - a);
    return (hi << a) | lo;
}
__host__ __device__ __forceinline__ uint32_t suffix_bucket(uint64_t pk, int s) {
    const uint32_t m = (1u << s) - 1u;
    return ((((uint32_t)(pk >> 32)) & m) << s) | ((uint32_t)pk & m);
}

// ---------------------------------------------------------------------------------------------------------
// database residency: SoA -> bucketed scan image (counting sort by bucket id)
// ---------------------------------------------------------------------------------------------------------
__global__ void k_check_counts(const uint64_t *__restrict__ targets, uint64_t n, uint32_t *__restrict__ counts, uint32_t *__restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = (uint32_t)(targets[i] >> 48);
    counts[i] = c;
    if (c == 0 || c > 32767) atomicAdd(bad, 1u);  // getCount is a signed short and must be > 0 (BlockManager.scala:232-234)
}

template <bool SUFFIX>
__global__ void k_image_hist(const uint64_t *__restrict__ targets, uint64_t n, Geometry geo, int width, uint32_t *__restrict__ bcount) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t pk = planar_key(targets[i], geo.c0, geo.lc);
    const uint32_t b = SUFFIX ? suffix_bucket(pk, width) : prefix_bucket(pk, geo.lc, width);
    atomicAdd(&bcount[b], 1u);
}

template <bool SUFFIX>
__global__ void k_image_scatter(const uint64_t *__restrict__ targets, uint64_t n, Geometry geo, int width,
                                const uint32_t *__restrict__ bstart, uint32_t *__restrict__ bfill, uint64_t *__restrict__ keys,
                                uint32_t *__restrict__ tidx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t pk = planar_key(targets[i], geo.c0, geo.lc);
    const uint32_t b = SUFFIX ? suffix_bucket(pk, width) : prefix_bucket(pk, geo.lc, width);
    const uint32_t pos = bstart[b] + atomicAdd(&bfill[b], 1u);
    keys[pos] = pk;
    tidx[pos] = (uint32_t)i;
}

// ---------------------------------------------------------------------------------------------------------
// candidate lists: every guide visits the buckets inside its Hamming ball (key ^ pattern)
// ---------------------------------------------------------------------------------------------------------
template <bool SUFFIX>
__global__ void k_guide_keys(const uint64_t *__restrict__ guides, uint32_t n, Geometry geo, int width, uint64_t *__restrict__ gkey,
                             uint32_t *__restrict__ gbucket, uint32_t *__restrict__ seg_begin /* nullable */, uint32_t *__restrict__ seg_end,
                             uint32_t *__restrict__ zero_buf, uint32_t n_zero) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t d = g; d < n_zero; d += gridDim.x * blockDim.x) zero_buf[d] = 0u;  // the partition histogram k_guide_part_hist adds into
    if (g >= n) return;
    if (seg_begin) { seg_begin[g] = 0u; seg_end[g] = 0u; }  // the hit segment of a guide without hits (k_segments only visits the others)
    const uint64_t pk = planar_key(guides[g], geo.c0, geo.lc);
    gkey[g] = pk;
    gbucket[g] = SUFFIX ? suffix_bucket(pk, width) : prefix_bucket(pk, geo.lc, width);
}

// Candidate lists in CSR form: for every bucket the ids of the guides whose Hamming ball reaches it (the 8-byte planar
// guide keys stay in a table small enough to live in L2 and are gathered by the compare kernel).  The
// (bucket, guide) entries are enumerated implicitly (bucket = guide bucket ^ pattern) and binned EXACTLY -- no
// capacity guess, so skewed guide sets (tiling libraries, repeats) cost nothing extra -- and without per-entry global
// atomics (device-scope atomics on random addresses run at ~1.3e10/s on MI355X: 4 ms for the 5.6e7 entries of the
// hg38-scale workload):
//   A1 k_guide_part_hist + k_part_sizes : the partition (= high bits of the bucket id) sizes, computed without touching
//                           the entries (XOR convolution of two histograms);  exclusive scan of the <= 4096 sizes;
//   A2 k_item_partition   : a block enumerates 256Ki entries, reserves one run per partition (one atomic each) and writes
//                           (low bucket bits, guide) records into the partition's exactly-sized staging range;
//   B  k_item_bin         : one block per partition counts its records per bucket in LDS, scans the counts, writes
//                           the CSR offsets of its buckets and scatters the guide ids into place (LDS atomics only).
constexpr int kPartThreads = 1024;
constexpr int kPartItemsPerBlock = 32768;
constexpr int kMaxPartBits = 12;   // <= 4096 partitions
constexpr int kMaxLowBits = 12;    // <= 4096 buckets per partition (11 bits preferred: see prepare_side)
constexpr int kBinStage = 14336;   // candidate ids staged in LDS per partition (56 KB: two blocks per CU); larger partitions scatter to memory
constexpr int kGidBits = 20;       // guides per batch < 2^20

struct ItemGeom {
    uint32_t n_guides, n_pat;
    uint32_t low_bits;       // bucket id = (partition << low_bits) | low
    uint32_t n_part;
    uint32_t item_base;      // first CSR slot of this image (the two images share the item array)
    uint64_t pat_magic;      // ceil(2^40 / n_pat) when n_pat < 2^18 (then x / n_pat == (x * pat_magic) >> 40 for x < 2^19), else 0
};

// entry number -> guide number inside a block's window: a 32-bit division costs ~30 instructions per entry and pass
__device__ __forceinline__ uint32_t div_pat(uint32_t x, const ItemGeom &ig) {
    return ig.pat_magic ? (uint32_t)(((uint64_t)x * ig.pat_magic) >> 40) : x / ig.n_pat;
}

// exclusive scan over the 1024 threads of a block (16 waves)
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *lds /* >= 16 */, uint32_t &total) {
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
    for (int w = 0; w < kPartThreads / 64; ++w) {
        const uint32_t s = lds[w];
        if ((uint32_t)w < wave) off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return off + incl - v;
}

// LDS counter index of partition / bucket x.  A wave's 64 entries are 64 patterns of one guide: their partitions differ from the
// guide's only in the few bits the patterns set, very often not in the five bits that select the LDS bank, and the counters of
// one wave then sit in one bank (SQ_LDS_BANK_CONFLICT = 96 % of the LDS cycles).  Folding the higher bits into the bank bits is a
// bijection on every power-of-two range >= 32 and spreads them.
__device__ __forceinline__ uint32_t lds_slot(uint32_t x) { return x ^ ((x >> 5) & 31u) ^ ((x >> 10) & 31u); }

// Partition sizes without enumerating the entries: bucket = guide bucket ^ pattern acts bit by bit, so the number of
// entries whose high bits equal q is  sum over patterns p of  #guides whose high bits equal q ^ high(p)  -- an XOR
// convolution of the guides' partition histogram with the patterns' (<= 4096 x n_pat additions instead of one pass
// over all n_guides x n_pat entries).
// a few blocks, each with an LDS histogram of its slice of the guides, merged with one global atomic per non-empty counter (100 000
// device-scope atomics on 2048 counters took 47 us; one block walking all guides took 30 us; eight blocks take a few).  ghist was
// cleared by k_guide_keys.
constexpr int kPartHistBlocks = 8;
__global__ __launch_bounds__(1024) void k_guide_part_hist(const uint32_t *__restrict__ gbucket, uint32_t n_guides, uint32_t low_bits, uint32_t n_part,
                                                          uint32_t *__restrict__ ghist, uint32_t *__restrict__ part_fill, uint32_t n_fill) {
    __shared__ uint32_t h[1 << kMaxPartBits];
    if (blockIdx.x == 0)
        for (uint32_t d = threadIdx.x; d < n_fill; d += blockDim.x) part_fill[d] = 0;  // the counters of the passes that follow (saves a fill launch)
    for (uint32_t d = threadIdx.x; d < n_part; d += blockDim.x) h[d] = 0;
    __syncthreads();
    const uint32_t per = (n_guides + gridDim.x - 1) / gridDim.x, g_end = min(n_guides, (blockIdx.x + 1) * per);
    for (uint32_t g = blockIdx.x * per + threadIdx.x; g < g_end; g += blockDim.x) atomicAdd(&h[lds_slot(gbucket[g] >> low_bits)], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < n_part; d += blockDim.x) {
        const uint32_t c = h[lds_slot(d)];
        if (c) atomicAdd(&ghist[d], c);
    }
}
// one wave per partition, the lanes stride over the patterns (a thread per partition left 4096 threads walking 529 patterns each: 36 us)
__global__ __launch_bounds__(256) void k_part_sizes(const uint32_t *__restrict__ ghist, const uint32_t *__restrict__ patterns, ItemGeom ig,
                                                    uint32_t *__restrict__ part_count) {
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= ig.n_part) return;
    uint32_t n = 0;
    for (uint32_t p = lane; p < ig.n_pat; p += 64) n += ghist[q ^ (patterns[p] >> ig.low_bits)];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
    if (lane == 0) part_count[q] = n;
}

template <bool WRITE>
__global__ __launch_bounds__(kPartThreads) void k_item_partition(const uint32_t *__restrict__ gbucket, const uint32_t *__restrict__ patterns, ItemGeom ig,
                                                                 const uint32_t *__restrict__ part_start, uint32_t *__restrict__ part_fill,
                                                                 uint32_t *__restrict__ part_items) {
    __shared__ uint32_t cur[1 << kMaxPartBits];
    const uint64_t total = (uint64_t)ig.n_guides * ig.n_pat;
    const uint64_t begin = (uint64_t)blockIdx.x * kPartItemsPerBlock;
    const uint64_t end = min(total, begin + (uint64_t)kPartItemsPerBlock);
    for (uint32_t d = threadIdx.x; d < ig.n_part; d += kPartThreads) cur[d] = 0;
    __syncthreads();
    // entry i = guide (i / n_pat), pattern (i % n_pat); 32-bit arithmetic relative to the block's first entry
    const uint32_t g_first = (uint32_t)(begin / ig.n_pat), j_first = (uint32_t)(begin - (uint64_t)g_first * ig.n_pat);
    const uint32_t n_here = (uint32_t)(end - begin);
    for (uint32_t o = threadIdx.x; o < n_here; o += kPartThreads) {
        const uint32_t x = j_first + o, q = div_pat(x, ig);
        const uint32_t b = gbucket[g_first + q] ^ patterns[x - q * ig.n_pat];
        atomicAdd(&cur[lds_slot(b >> ig.low_bits)], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < ig.n_part; d += kPartThreads) {
        const uint32_t c = cur[lds_slot(d)];
        if (!WRITE) { if (c) atomicAdd(&part_fill[d], c); }                              // A1: partition sizes
        else cur[lds_slot(d)] = c ? part_start[d] + atomicAdd(&part_fill[d], c) : 0u;  // A2: start of this block's run
    }
    if (!WRITE) return;
    __syncthreads();
    for (uint32_t o = threadIdx.x; o < n_here; o += kPartThreads) {
        const uint32_t x = j_first + o, q = div_pat(x, ig), g = g_first + q;
        const uint32_t b = gbucket[g] ^ patterns[x - q * ig.n_pat];
        const uint32_t pos = atomicAdd(&cur[lds_slot(b >> ig.low_bits)], 1u);
        part_items[pos] = ((b & ((1u << ig.low_bits) - 1u)) << kGidBits) | g;
    }
}

// One block per partition.  The partition's candidate ids are put in bucket order inside LDS and leave as one
// contiguous, coalesced copy: scattering 4-byte stores straight to memory costs a partial-line write-back each once the
// concurrently open output windows exceed the L2 (1.1 ms for the 5.3e7 entries of the hg38-scale prefix image).
__global__ __launch_bounds__(kPartThreads, 8) void k_item_bin(const uint32_t *__restrict__ part_start, const uint32_t *__restrict__ part_items, ItemGeom ig,
                                                           uint32_t *__restrict__ istart, uint32_t *__restrict__ item_gid) {
    __shared__ uint32_t cnt[1 << kMaxLowBits];
    __shared__ uint32_t stage[kBinStage];
    __shared__ uint32_t scan_lds[16];
    const uint32_t d = blockIdx.x, nlow = 1u << ig.low_bits;
    for (uint32_t l = threadIdx.x; l < nlow; l += kPartThreads) cnt[l] = 0;
    __syncthreads();
    const uint32_t p0 = part_start[d], n = part_start[d + 1] - p0;
    const uint32_t *__restrict__ src = part_items + p0;
    const uint32_t gbase = ig.item_base + p0;  // records of partition d occupy CSR slots [item_base + p0, + n)
    const uint32_t per = (nlow + kPartThreads - 1) / kPartThreads, l0 = threadIdx.x * per;
    // exclusive scan of the nlow counters (every thread owns nlow / 1024 consecutive ones, at most 4): CSR offsets out, counters
    // become scatter cursors
    auto scan_counters = [&]() {
        uint32_t mine = 0;
        for (uint32_t k = 0; k < per; ++k)
            if (l0 + k < nlow) mine += cnt[lds_slot(l0 + k)];
        uint32_t tot;
        uint32_t off = block_exclusive_scan_1024(mine, scan_lds, tot);
        for (uint32_t k = 0; k < per; ++k)
            if (l0 + k < nlow) {
                const uint32_t c = cnt[lds_slot(l0 + k)];
                istart[((uint64_t)d << ig.low_bits) + l0 + k] = gbase + off;
                cnt[lds_slot(l0 + k)] = off;
                off += c;
            }
        if (d == gridDim.x - 1 && threadIdx.x == kPartThreads - 1) istart[(uint64_t)ig.n_part << ig.low_bits] = gbase + n;
    };
    if (n <= (uint32_t)kBinStage) {
        // the usual case: the whole partition sits in registers (14 records per thread, all loads in flight at once), is counted and
        // placed from there -- one read of the records instead of two, one memory round trip instead of eight
        constexpr int kR = kBinStage / kPartThreads;
        static_assert(kR * kPartThreads == kBinStage, "kBinStage is a multiple of the block size");
        uint32_t r[kR];
#pragma unroll
        for (int u = 0; u < kR; ++u) { const uint32_t k = u * kPartThreads + threadIdx.x; r[u] = k < n ? src[k] : 0xFFFFFFFFu; }
#pragma unroll
        for (int u = 0; u < kR; ++u) if ((uint32_t)(u * kPartThreads) + threadIdx.x < n) atomicAdd(&cnt[lds_slot(r[u] >> kGidBits)], 1u);
        __syncthreads();
        scan_counters();
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kR; ++u)
            if ((uint32_t)(u * kPartThreads) + threadIdx.x < n) stage[atomicAdd(&cnt[lds_slot(r[u] >> kGidBits)], 1u)] = r[u] & ((1u << kGidBits) - 1u);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += kPartThreads) item_gid[gbase + k] = stage[k];
        return;
    }
    // oversized partition (skewed guide sets): two passes over the records, scattered stores
    constexpr int kU = 4;  // records in flight per thread
    for (uint32_t k0 = 0; k0 < n; k0 += kPartThreads * kU) {
        uint32_t r[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { const uint32_t k = k0 + u * kPartThreads + threadIdx.x; r[u] = k < n ? src[k] : 0xFFFFFFFFu; }
#pragma unroll
        for (int u = 0; u < kU; ++u) if (k0 + u * kPartThreads + threadIdx.x < n) atomicAdd(&cnt[lds_slot(r[u] >> kGidBits)], 1u);
    }
    __syncthreads();
    scan_counters();
    __syncthreads();
    for (uint32_t k0 = 0; k0 < n; k0 += kPartThreads * kU) {
        uint32_t r[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { const uint32_t k = k0 + u * kPartThreads + threadIdx.x; r[u] = k < n ? src[k] : 0xFFFFFFFFu; }
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (k0 + u * kPartThreads + threadIdx.x < n) item_gid[gbase + atomicAdd(&cnt[lds_slot(r[u] >> kGidBits)], 1u)] = r[u] & ((1u << kGidBits) - 1u);
    }
}

constexpr int kTileTargets = 256;  // targets per work item (a bucket, or a 256-target slice of a large bucket)
constexpr int kTileChunks = kTileTargets / 64;

__global__ void k_tile_count(const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ istart, uint32_t n_buckets,
                             uint32_t *__restrict__ tcount) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    const uint32_t nt = bstart[b + 1] - bstart[b];
    tcount[b] = (istart[b + 1] != istart[b]) ? (nt + kTileTargets - 1) / kTileTargets : 0;
}

// work item ("tile") = {first key, #keys | side << 31, first candidate, #candidates}; the items of both images share
// one list so that ONE compare launch covers the prefix and the suffix pass
__global__ void k_tile_fill(const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ istart, const uint32_t *__restrict__ tstart,
                            uint32_t n_buckets, const uint32_t *__restrict__ tile_base, uint32_t side, uint4 *__restrict__ tiles) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    const uint32_t t0 = tstart[b], nt = tstart[b + 1] - t0;
    if (!nt) return;
    const uint32_t k0 = bstart[b], kn = bstart[b + 1] - k0, g0 = istart[b], gn = istart[b + 1] - g0;
    uint4 *out = tiles + *tile_base + t0;
    for (uint32_t c = 0; c < nt; ++c) {
        const uint32_t kb = c * kTileTargets;
        out[c] = make_uint4(k0 + kb, min(kn - kb, (uint32_t)kTileTargets) | (side << 31), g0, gn);
    }
}

// ---------------------------------------------------------------------------------------------------------
// THE HOT KERNEL.  One wave owns one work item at a time: <= 256 targets of one bucket (streamed coalesced from the
// scan image) against that bucket's candidate guides.  The kernel is latency-, not bandwidth-bound unless every
// load is issued a whole item ahead (HBM round trips under load are ~10x the compute of one item), hence the
// three-deep software pipeline:
//      item i+3: descriptor (scalar load)           item i+2: slot row = candidate guide ids (one coalesced load)
//      item i+1: planar guide keys (gather from the L2-resident table) + all target keys (<= 4 coalesced loads)
//      item i  : computed entirely out of the wave's LDS strip -- no global access in the loops at all.
//   * candidates are read back as LDS broadcasts: 4 VALU per (guide, 64 targets): v_xor, v_bitop3 (xor|or), v_bcnt,
//     v_cmp -- no readlane, no memory wait;
//   * short chunks (the <= 32 / 16 / 8 target tail of a bucket) are replicated 2 / 4 / 8 times across the wave and
//     tested against 2 / 4 / 8 different guides per step, so the tail does not waste the lanes;
//   * a hit is staged as (guide, side, position in the image), so the hot loop never waits on memory even when it hits;
//   * hits are compacted with ballot + mbcnt into a per-wave LDS staging buffer and flushed with ONE global atomic
//     per ~200 hits (a single global cursor saturates far below the hit rate); the flush looks the database index up and
//     writes the final sort key (guide << tbits) | index.
//   Suffix-image items: the same pair can only also be found through the prefix image when its prefix part has
//   <= r1 mismatches, so it is emitted from a suffix item only if the prefix part has MORE than r1.
// ---------------------------------------------------------------------------------------------------------
constexpr int kCmpThreads = 256;
// staged hits per wave.  Every flush is one atomic on the one global hit cursor, and same-address atomics complete at ~88 per
// microsecond on this part: at 5 mismatches (1.1e8 hits) the launch time WAS the flush count (11.6 ms with 192 entries, 7.8 ms with
// 280, 7.6 ms with 360).  360 entries leave room for seven blocks per CU (4 x 360 x 8 B + 11 KB of keys each, 160 KB of LDS);
// same-box sweep at 4 / 5 mismatches: 192 -> 1.90 / 11.6 ms, 280 (eight blocks) -> 1.87 / 7.9, 360 -> 1.86 / 7.6, 480 (six blocks)
// -> 1.98 / 8.0, 640 (five) -> 2.19 / 8.8.
constexpr int kStage = 360;

constexpr uint32_t kPairSlotBase = 16, kPairSlots = 64;  // pair counters live at cursor[16 .. 16 + 2 * 64)
constexpr uint32_t kTileStatBase = 13;  // cursor[13], cursor[14]: work items of the prefix / suffix image in this launch
constexpr uint32_t kFlushArgBase = kPairSlotBase + 2 * kPairSlots;  // FlushArgs of the current launch live behind them
// What a wave needs only when it empties its staged hits (once per ~130 hits).  It is read from device memory at that point
// instead of being kernel arguments: as arguments the twelve scalars stay live through the whole hot loop, and the register
// allocator pays for them by spilling the item descriptors of the software pipeline to VGPR lanes on every item.
struct FlushArgs {
    uint64_t *hits;
    uint64_t cap;
    const uint32_t *tidx_p, *tidx_s;
    uint32_t guide_base;  // first guide of this batch
    int tbits;            // hit key = (global guide << tbits) | database index
};
static_assert(sizeof(FlushArgs) == 40, "FlushArgs is stored as five 64-bit words");
struct CompareArgs {
    const uint4 *tiles;
    const uint32_t *n_tiles_a, *n_tiles_b;
    const uint64_t *keys[2];
    const uint32_t *slots;
    const uint64_t *gkey;
    int max_mm;
    uint32_t prefix_mask;
    int r1;
    unsigned long long *cursor;  // [0] hit cursor; [kPairSlotBase + 2 * slot + side] executed pair tests, summed by the host; [kFlushArgBase ..] FlushArgs
};

// before every compare launch: clears the per-launch pair counters and stores the launch's FlushArgs (one launch in place of a memset)
__global__ void k_compare_setup(unsigned long long *__restrict__ cursor, FlushArgs f, int first_batch) {
    if (first_batch && threadIdx.x < 8) cursor[threadIdx.x] = 0ull;  // hit cursor and the zero word: once per scan, they run across batches
    if (threadIdx.x < 2 * kPairSlots) cursor[kPairSlotBase + threadIdx.x] = 0ull;
    if (threadIdx.x == 0) *reinterpret_cast<FlushArgs *>(cursor + kFlushArgBase) = f;
}

struct HitStage {
    uint64_t *my;
    uint32_t fill;
    uint32_t lane;
    unsigned long long *cursor;

    __device__ __forceinline__ void flush() {
        // five wave-uniform words, moved to scalar registers so they do not raise the vector register count of the hot loop
        const unsigned long long *q = cursor + kFlushArgBase;
        auto uni = [](unsigned long long v) -> unsigned long long {
            return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);  // the builtin returns int
        };
        FlushArgs f;
        f.hits = (uint64_t *)uni(q[0]); f.cap = uni(q[1]); f.tidx_p = (const uint32_t *)uni(q[2]); f.tidx_s = (const uint32_t *)uni(q[3]);
        const unsigned long long gb = uni(q[4]);
        f.guide_base = (uint32_t)gb; f.tbits = (int)(gb >> 32);
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(cursor, (unsigned long long)fill);
        base = __shfl(base, 0, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // staged record = (batch-local guide << 32) | side << 31 | position in that side's image; it leaves as the sort key
        // (global guide << tbits) | database index -- the lookup rides on the flush instead of a pass of its own over all hits
        for (uint32_t i = lane; i < fill; i += 64)
            if (base + i < f.cap) {
                const uint64_t h = my[i];
                const uint32_t lo = (uint32_t)h, pos = lo & 0x7FFFFFFFu;
                const uint32_t ti = (lo >> 31) ? f.tidx_s[pos] : f.tidx_p[pos];
                f.hits[base + i] = ((uint64_t)((uint32_t)(h >> 32) + f.guide_base) << f.tbits) | ti;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        fill = 0;
    }
    // wave-uniform call: `mask` = ballot of `hit`; every hitting lane records (candidate guide id, position)
    __device__ __forceinline__ void push(uint64_t mask, bool hit, const uint32_t *gid_lds, uint32_t idx, uint32_t pos) {
        if (hit) my[fill + mbcnt(mask)] = ((uint64_t)gid_lds[idx] << 32) | pos;
        fill += (uint32_t)__popcll(mask);
        if (fill > kStage - 64) flush();  // always leave room for one more wave-wide batch (the predicate never lives across the flush)
    }
};

struct WaveCtx {  // per-wave state shared by the chunk loops
    const CompareArgs *a;
    HitStage *hs;
    const uint64_t *key_lds;  // the item's target keys
    const uint64_t *gk_lds;   // 64 candidate keys (sentinel beyond n)
    const uint32_t *gid_lds;  // 64 candidate guide ids
    uint32_t lane;
    uint32_t side_bit;         // side << 31, or'ed into the recorded position
    uint32_t pm;               // suffix item: the prefix part of the comparison mask; prefix item: 0
    int r1s;                   // suffix item: r1; prefix item: -1
};

// one 64-lane step against W guides at once; `cnt` <= 64 / W targets, key_lds[koff ..], image position pos0 + ..
// CHECK: max_mm is too large for the sentinel to be safe, test the candidate index explicitly
// FULL: all 64 lanes have a target (W == 1 only): no validity mask, no substitute key
template <int W, bool CHECK, bool FULL = false>
__device__ __forceinline__ void scan_chunk(const WaveCtx &w, uint32_t koff, uint32_t pos0, uint32_t cnt, uint32_t n) {
    static_assert(!FULL || W == 1, "a full chunk is 64 targets wide");
    constexpr uint32_t SUB = 64 / W;
    const uint32_t tl = w.lane & (SUB - 1), gs = w.lane / SUB;
    const bool valid = FULL ? true : tl < cnt;
    // lanes without a target carry the all-ones key: it differs from every candidate (real or sentinel) in >= 12 bits
    const uint64_t k = valid ? w.key_lds[koff + tl] : ~0ull;
    const uint32_t kh = (uint32_t)(k >> 32), kl = (uint32_t)k;
    const uint32_t max_mm = (uint32_t)w.a->max_mm;
    const uint32_t iters = (n + W - 1) / W;
    // the hit predicate stays a lane mask from the compare to the staged store: a candidate without a hit costs one vector
    // compare and one scalar branch
    auto report = [&](uint32_t p, uint32_t y, uint32_t idx) {
        bool h = p <= max_mm;
        if (CHECK) h = h && valid && idx < n;  // the sentinels are not safe for max_mm >= 12
        const uint64_t m1 = __builtin_amdgcn_ballot_w64(h);
        if (!m1) return;
        // suffix items keep a pair only if its prefix part has more than r1 mismatches; prefix items run the same three
        // instructions with pm = 0, r1s = -1 (always true) instead of a branch that would park the predicate in a VGPR
        const bool far = (int)__popc(y & w.pm) > w.r1s;
        const uint64_t m = m1 & __builtin_amdgcn_ballot_w64(far);  // the two lane masks meet on the scalar side
        if (m) w.hs->push(m, h && far, w.gid_lds, idx, (pos0 + tl) | w.side_bit);
    };
    // four candidates per step; ONE vector compare and ONE scalar branch decide whether any of the 256 pairs is within
    // max_mm (the scalar unit is shared by the CU's four SIMDs: mask algebra per pair would make it the bottleneck)
    // the loop variable is the LDS address of the group's first candidate key (one add, one compare, one branch of loop control
    // per group of four); the candidate index is only derived from it when something hit
    typedef __attribute__((address_space(3))) const uint64_t lds_key;  // 32-bit LDS addresses: the loop control stays in single registers
    auto dist_at = [&](lds_key *gp, uint32_t &y) -> uint32_t {
        const uint64_t g = *gp;
        y = __builtin_amdgcn_bitop3_b32((uint32_t)g, kh ^ (uint32_t)(g >> 32), kl, 0xde);  // (gl ^ kl) | (gh ^ kh)
        return (uint32_t)__popc(y);
    };
    if constexpr (W == 1) {
        // full-width chunks (all of the suffix image, the first chunk of most prefix buckets): the candidate address is wave-uniform,
        // so it is the loop variable itself -- one add, one compare, one branch of loop control per group of four
        lds_key *const gbase = (lds_key *)w.gk_lds;
        lds_key *gp = gbase, *const g4 = gbase + (iters & ~3u), *const ge = gbase + iters;
        for (; gp != g4; gp += 4) {
            uint32_t y0, y1, y2, y3;
            const uint32_t p0 = dist_at(gp, y0), p1 = dist_at(gp + 1, y1), p2 = dist_at(gp + 2, y2), p3 = dist_at(gp + 3, y3);
            const uint32_t best = min(min(p0, p1), min(p2, p3));
            if (__builtin_amdgcn_ballot_w64(best <= max_mm)) {
                const uint32_t i0 = (uint32_t)(gp - gbase);
                report(p0, y0, i0);
                report(p1, y1, i0 + 1);
                report(p2, y2, i0 + 2);
                report(p3, y3, i0 + 3);
            }
        }
        for (; gp != ge; ++gp) {
            uint32_t y;
            const uint32_t p = dist_at(gp, y);
            if (__builtin_amdgcn_ballot_w64(p <= max_mm)) report(p, y, (uint32_t)(gp - gbase));
        }
    } else {
        lds_key *const gbase = (lds_key *)w.gk_lds;
        uint32_t j = 0;
        for (; j + 4 <= iters; j += 4) {
            uint32_t y0, y1, y2, y3;
            const uint32_t i0 = j * W + gs, i1 = i0 + W, i2 = i0 + 2 * W, i3 = i0 + 3 * W;
            const uint32_t p0 = dist_at(gbase + i0, y0), p1 = dist_at(gbase + i1, y1), p2 = dist_at(gbase + i2, y2), p3 = dist_at(gbase + i3, y3);
            const uint32_t best = min(min(p0, p1), min(p2, p3));
            if (__builtin_amdgcn_ballot_w64(best <= max_mm)) {
                report(p0, y0, i0);
                report(p1, y1, i1);
                report(p2, y2, i2);
                report(p3, y3, i3);
            }
        }
        for (; j < iters; ++j) {
            uint32_t y;
            const uint32_t i0 = j * W + gs;
            const uint32_t p = dist_at(gbase + i0, y);
            if (__builtin_amdgcn_ballot_w64(p <= max_mm)) report(p, y, i0);
        }
    }
}

// the read-only streams are separate __restrict__ kernel parameters (noalias lets the compiler keep the descriptor
// loads on the scalar unit and reorder the vector loads around the hit stores)
template <bool CHECK>
__global__ __launch_bounds__(kCmpThreads, 7) void k_compare(const uint4 *__restrict__ tiles, const uint64_t *__restrict__ keys_p,
                                                         const uint64_t *__restrict__ keys_s, const uint32_t *__restrict__ slots,
                                                         const uint64_t *__restrict__ gkey, const CompareArgs a) {
    __shared__ uint64_t stage[kCmpThreads / 64][kStage];
    __shared__ uint64_t key_lds[kCmpThreads / 64][kTileTargets];
    __shared__ uint64_t gk_lds[kCmpThreads / 64][64];
    __shared__ uint32_t gid_lds[kCmpThreads / 64][64];
    __shared__ unsigned long long blk_pairs[2];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (kCmpThreads / 64);
    const uint32_t n_tiles = *a.n_tiles_a + *a.n_tiles_b;
    if (threadIdx.x < 2) blk_pairs[threadIdx.x] = 0;
    // the work-item counts ride back to the host in the counter block (one copy after the launch instead of three)
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.cursor[kTileStatBase] = *a.n_tiles_a; a.cursor[kTileStatBase + 1] = *a.n_tiles_b; }
    __syncthreads();
    HitStage hs{stage[wave], 0u, lane, a.cursor};
    unsigned long long pairs[2] = {0, 0};
    // padding candidate: the 12 unused high bits of both planes set, the 20 used ones clear.  It differs from every real
    // key in those 12 bits and from the all-ones key of an idle lane in the 20 low ones: never within max_mm < 12
    const uint64_t sentinel = 0xFFF00000FFF00000ull;
    WaveCtx w{&a, &hs, key_lds[wave], gk_lds[wave], gid_lds[wave], lane, 0u, 0u, -1};

    uint32_t t = blockIdx.x * (kCmpThreads / 64) + wave;
    if (t >= n_tiles) goto done;
    {
        // every load below is unconditional with a clamped index (straight-line code lets the compiler count
        // outstanding loads exactly instead of draining the queue)
        // descriptors past the end are replaced by the last one (its loads are harmless and never consumed)
        auto load_desc = [&](uint32_t ti) -> uint4 { return tiles[min(ti, n_tiles - 1)]; };
        // uniform base pointer + small per-lane index: the loads use the scalar-base addressing form
        auto load_gid = [&](const uint4 &d) -> uint32_t { return (slots + d.z)[min(lane, max(d.w, 1u) - 1u)]; };
        auto load_key = [&](const uint4 &d, uint32_t c) -> uint64_t {
            const uint32_t kc = d.y & 0x7FFFFFFFu;
            const uint64_t *__restrict__ kp = ((d.y >> 31) ? keys_s : keys_p) + d.x;
            return kp[min(c * 64u + lane, max(kc, 1u) - 1u)];
        };
        // prologue
        uint4 d0 = load_desc(t), d1 = load_desc(t + n_waves), d2 = load_desc(t + 2 * n_waves);
        uint32_t gid0 = load_gid(d0), gid1 = load_gid(d1);
        uint64_t gk0 = gkey[gid0];
        uint64_t k0[kTileChunks];
#pragma unroll
        for (int c = 0; c < kTileChunks; ++c) k0[c] = load_key(d0, c);

        while (t < n_tiles) {
            const uint4 cur = d0;
            const uint32_t side = cur.y >> 31;  // wave-uniform
            const uint32_t kcnt = cur.y & 0x7FFFFFFFu, ng = cur.w;
            // ---- stage A: park item i in the LDS strip (waits for everything requested one item ago) ----
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            gk_lds[wave][lane] = (lane < ng) ? gk0 : sentinel;
            gid_lds[wave][lane] = gid0;
#pragma unroll
            for (int c = 0; c < kTileChunks; ++c) key_lds[wave][c * 64 + lane] = k0[c];
            // ---- stage B: request item i+1 (guide keys, target keys), i+2 (slot row), i+3 (descriptor) ----
            const uint4 d3 = load_desc(t + 3 * n_waves);
            const uint32_t gid2 = load_gid(d2);
            gk0 = gkey[gid1];
#pragma unroll
            for (int c = 0; c < kTileChunks; ++c) k0[c] = load_key(d1, c);
            gid0 = gid1; gid1 = gid2;
            d0 = d1; d1 = d2; d2 = d3;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- stage C: compute item i out of LDS ----
            w.pm = side ? a.prefix_mask : 0u;
            w.r1s = side ? a.r1 : -1;
            w.side_bit = side << 31;
            pairs[side] += (unsigned long long)kcnt * ng;
            for (uint32_t g0 = 0; g0 < ng; g0 += 64) {
                if (g0) {  // rows longer than 64 candidates (rare): fetch the next 64 in place
                    const uint32_t gi = slots[cur.z + min(g0 + lane, ng - 1)];
                    const uint64_t gk = gkey[gi];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    gk_lds[wave][lane] = (g0 + lane < ng) ? gk : sentinel;
                    gid_lds[wave][lane] = gi;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                const uint32_t n = min(ng - g0, 64u);
                for (uint32_t c = 0; c < kcnt; c += 64) {
                    const uint32_t cnt = min(kcnt - c, 64u);
                    if (cnt == 64) {
                        scan_chunk<1, CHECK, true>(w, c, cur.x + c, cnt, n);
                    } else {
                        // the tail of a bucket: the candidate count goes through an opaque register so that the trip counts of the
                        // four packings are worked out here, for the one that runs, and not hoisted in front of every item's chunk loop
                        uint32_t nt = n;
                        asm volatile("" : "+s"(nt));
                        if (cnt > 32) scan_chunk<1, CHECK>(w, c, cur.x + c, cnt, nt);
                        else if (cnt > 16) scan_chunk<2, CHECK>(w, c, cur.x + c, cnt, nt);
                        else if (cnt > 8) scan_chunk<4, CHECK>(w, c, cur.x + c, cnt, nt);
                        else scan_chunk<8, CHECK>(w, c, cur.x + c, cnt, nt);
                    }
                }
            }
            t += n_waves;
        }
    }
done:
    if (hs.fill) hs.flush();
    if (lane == 0) {
        if (pairs[0]) atomicAdd(&blk_pairs[0], pairs[0]);
        if (pairs[1]) atomicAdd(&blk_pairs[1], pairs[1]);
    }
    __syncthreads();
    // 64 x 2 counters instead of 2: atomics on ONE address complete at ~90 per microsecond, and 16 384 blocks reporting to the
    // same two words held every launch for ~0.2 ms after its last block had finished (visible as the whole compare time of small calls)
    if (threadIdx.x < 2 && blk_pairs[threadIdx.x]) atomicAdd(a.cursor + kPairSlotBase + (blockIdx.x & (kPairSlots - 1)) * 2 + threadIdx.x, blk_pairs[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------
// epilogue: ordered cut-off (crispr/CRISPRSiteOT.scala:39-46) + per-hit scores + per-guide aggregates.
// The hits are sorted by (guide, database index); one WAVE works on one guide's segment.
// ---------------------------------------------------------------------------------------------------------
__global__ void k_segments(const uint64_t *__restrict__ hits, uint64_t n, int tbits, uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = (uint32_t)(hits[i] >> tbits);
    if (i == 0 || (uint32_t)(hits[i - 1] >> tbits) != g) seg_begin[g] = (uint32_t)i;
    if (i == n - 1 || (uint32_t)(hits[i + 1] >> tbits) != g) seg_end[g] = (uint32_t)(i + 1);
}

// the target long of every raw hit, in sorted order (the one random gather of the epilogue)
__global__ void k_hit_targets(const uint64_t *__restrict__ hits, uint64_t n, int tbits, const uint64_t *__restrict__ targets, uint64_t *__restrict__ st) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st[i] = targets[hits[i] & ((1ull << tbits) - 1ull)];
}

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ double bcast_f64(double v, uint32_t l) {  // lane l's value in every lane (through SGPRs)
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)u, l), hi = __builtin_amdgcn_readlane((uint32_t)(u >> 32), l);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor(v, d, 64));
    return v;
}

// a hit is kept iff the running position total BEFORE it is < overflow; the total grows by the hit's position count.
// Counts are >= 1, so the kept hits of a 64-hit chunk are a prefix of it.  With totals != nullptr the kernel instead
// reports min(sum of all counts, overflow) (multi-GPU exchange).
__global__ __launch_bounds__(256) void k_cutoff(const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_end, const uint64_t *__restrict__ st,
                                                const uint32_t *__restrict__ prior, uint32_t n_guides, uint32_t overflow, uint32_t *__restrict__ n_ret,
                                                uint32_t *__restrict__ ot_count, uint32_t *__restrict__ full, uint32_t *__restrict__ totals) {
    const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_guides) return;
    const uint32_t b = seg_begin[g], e = seg_end[g], p0 = prior ? prior[g] : 0u;
    uint32_t run = p0, kept = 0;
    for (uint32_t i = b; i < e && run < overflow; i += 64) {
        const bool in = i + lane < e;
        const uint32_t c = in ? (uint32_t)(st[i + lane] >> 48) : 0u;
        const uint32_t incl = wave_inclusive_scan_u32(c, lane);
        const bool keep = in && (run + (incl - c) < overflow);
        const uint32_t nk = (uint32_t)__popcll(__ballot(keep));
        kept += nk;
        if (nk) run += __shfl(incl, nk - 1, 64);
    }
    if (lane == 0) {
        if (totals) totals[g] = min(run, overflow);
        else { n_ret[g] = kept; ot_count[g] = run - p0; full[g] = run >= overflow; }
    }
}

struct ScoreTables {
    double cfd_mm[20 * 4 * 4];
    double cfd_pam[16];
    double hsu_coeff[20];
    double jost[19 * 4 * 4];  // [position - 1][off-target base][guide base], 1.0 on the diagonal
};

// mismatches + pam*CFD (Doench2016CFDScore.scala:67-73,132-151) + Hsu2013 hit score (CrisprMitEduOffTarget.scala:107-148)
// of one (guide, target) pair; both scores are NaN for a 0-mismatch hit (the on-target itself) and for enzymes the
// models are not defined over.  The multiplications run in the reference's order (position 0..19, PAM last).  The reference
// multiplies through all twenty positions; where the bases agree its factor is exactly 1.0 (Doench) or absent (Hsu), and x * 1.0
// is x bit for bit, so only the mismatching positions -- the set bits of the planar XOR, highest bit = base 0 -- are visited:
// at most maxMismatch iterations instead of twenty.
__device__ __forceinline__ void score_pair(uint64_t gd, uint64_t t, const Geometry &geo, const ScoreTables *__restrict__ tab, int &mm_out, double &cfd,
                                           double &hsu) {
    const uint64_t pg = planar_key(gd, geo.c0, geo.lc), pt = planar_key(t, geo.c0, geo.lc);
    const uint32_t y = ((uint32_t)(pg >> 32) ^ (uint32_t)(pt >> 32)) | ((uint32_t)pg ^ (uint32_t)pt);
    const int mm = __popc(y);
    mm_out = mm;
    cfd = __builtin_nan("");
    hsu = __builtin_nan("");
    if (geo.cas9_23 && mm != 0) {
        // base i (0 = 5' end) of a 23-mer sits at bits [2(22-i)+1 : 2(22-i)] of the long and at bit 19-i of the planes
        double score = 1.0, part_one = 1.0;
        for (uint32_t m = y; m;) {
            const int hb = 31 - __clz((int)m);
            m ^= 1u << hb;
            const int b = 19 - hb, sh = 2 * (22 - b);
            const uint32_t gb = (uint32_t)(gd >> sh) & 3u, ob = (uint32_t)(t >> sh) & 3u;
            score *= tab->cfd_mm[b * 16 + gb * 4 + ob];
            part_one = part_one * (1.0 - tab->hsu_coeff[b]);
        }
        const int first = 19 - (31 - __clz((int)y)), last = 19 - (__ffs((int)y) - 1);
        cfd = tab->cfd_pam[(uint32_t)t & 15u] * score;
        double part_two = 1.0;
        if (mm >= 2) {
            const double avg = (double)(last - first) / (double)(mm - 1);
            part_two = 1.0 / ((((19 - avg) / 19.0) * 4.0) + 1.0);
        }
        const double part_three = 1.0 / (double)(mm * mm);
        const double total = part_one * part_two * part_three * 100.0;
        const uint32_t p21 = ((uint32_t)t >> 2) & 3u, p22 = (uint32_t)t & 3u;  // A C G T = 0 1 2 3
        double adj = 0.01;
        if (p22 == 2u) adj = p21 == 2u ? 1.0 : p21 == 0u ? 0.26 : p21 == 1u ? 0.11 : 0.01;
        hsu = total * adj;
    }
}

// Jost & Santos CRISPRi activity of one (guide, off-target) pair, JostAndSantosCRISPRi.calc_score :92-127: the product over
// the mismatching positions 1..19 of the mean activity for (position, off-target base, complement of the guide base),
// multiplied in ascending position.  20-mers (scan length 23) skip their first base, 19-mers (22) use all of theirs.
// Defined for every Cas9 pack (:53-58); the caller skips pairs with no mismatch among the compared bases (:40).
// Agreeing positions carry 1.0 in the table, so again only the mismatching ones are visited.
__device__ __forceinline__ double jost_pair(uint64_t gd, uint64_t t, const Geometry &geo, const ScoreTables *__restrict__ tab) {
    const uint64_t pg = planar_key(gd, geo.c0, geo.lc), pt = planar_key(t, geo.c0, geo.lc);
    const uint32_t y = ((uint32_t)(pg >> 32) ^ (uint32_t)(pt >> 32)) | ((uint32_t)pg ^ (uint32_t)pt);
    const int first = geo.scan_len == 23 ? 1 : 0;
    double total = 1.0;
    for (uint32_t m = y; m;) {
        const int hb = 31 - __clz((int)m);
        m ^= 1u << hb;
        const int b = geo.lc - 1 - hb, k = b - first;  // base b of the protospacer = table row k
        if (k < 0 || k >= 19) continue;
        const int sh = 2 * (geo.scan_len - 1 - b);
        const uint32_t gb = (uint32_t)(gd >> sh) & 3u, ob = (uint32_t)(t >> sh) & 3u;
        total *= tab->jost[k * 16 + ob * 4 + gb];
    }
    return total;
}

// per retained hit of a discover scan: target long, mismatches, position count, database index and the two scores
__global__ void k_score_hits(const uint64_t *__restrict__ hits, uint64_t n_hits, int tbits, const uint32_t *__restrict__ seg_begin,
                             const uint32_t *__restrict__ n_ret, const uint64_t *__restrict__ ret_off, const uint64_t *__restrict__ st,
                             const uint64_t *__restrict__ guides, Geometry geo, const ScoreTables *__restrict__ tab, uint64_t *__restrict__ out_target,
                             uint8_t *__restrict__ out_mm, uint32_t *__restrict__ out_cnt, uint32_t *__restrict__ out_tidx,
                             double *__restrict__ out_cfd, double *__restrict__ out_hsu, double *__restrict__ out_jost /* may be null */) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    const uint64_t key = hits[i];
    const uint32_t g = (uint32_t)(key >> tbits), ti = (uint32_t)(key & ((1ull << tbits) - 1ull));
    const uint32_t local = (uint32_t)i - seg_begin[g];
    if (local >= n_ret[g]) return;
    const uint64_t o = ret_off[g] + local;
    const uint64_t t = st[i];
    int mm;
    double cfd, hsu;
    score_pair(guides[g], t, geo, tab, mm, cfd, hsu);
    out_target[o] = t;
    out_mm[o] = (uint8_t)mm;
    out_cnt[o] = (uint32_t)(t >> 48);
    out_tidx[o] = ti;
    out_cfd[o] = cfd;
    out_hsu[o] = hsu;
    if (out_jost) out_jost[o] = (geo.c0 == 3 && mm != 0) ? jost_pair(guides[g], t, geo, tab) : __builtin_nan("");
}

// the same for caller-supplied hit lists (the `score` path: hit lists re-read from a discover table)
__global__ void k_score_list(const uint64_t *__restrict__ hit_targets, const uint32_t *__restrict__ hit_guide, uint64_t n_hits,
                             const uint64_t *__restrict__ guides, Geometry geo, const ScoreTables *__restrict__ tab, uint8_t *__restrict__ out_mm,
                             uint32_t *__restrict__ out_cnt, double *__restrict__ out_cfd, double *__restrict__ out_hsu, double *__restrict__ out_jost) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    const uint64_t t = hit_targets[i];
    int mm;
    double cfd, hsu;
    score_pair(guides[hit_guide[i]], t, geo, tab, mm, cfd, hsu);
    out_mm[i] = (uint8_t)mm;
    out_cnt[i] = (uint32_t)(t >> 48);
    out_cfd[i] = cfd;
    out_hsu[i] = hsu;
    if (out_jost) out_jost[i] = (geo.c0 == 3 && mm != 0) ? jost_pair(guides[hit_guide[i]], t, geo, tab) : __builtin_nan("");
}

struct GuideSummary {  // mirrors ffh_guide_summary
    uint32_t n_hits, ot_count, overflow, hist[5], closest, closest_count, in_genome, n_scored;
    double cfd_max, cfd_sum, hsu_sum, jost_max, jost_sum;
};

// The ordered f64 walk shared by the two aggregation kernels.  A wave parks its lanes' addends in LDS and every lane then folds
// them in lane order -- the same sequence of additions as a scalar loop over the hits in database order, hence bit-identical sums
// -- reading each addend pair back with ONE uniform LDS read.  (Broadcasting with v_readlane cost four scalar-register moves per
// hit on the vector pipe; the adds themselves are two.)  Lanes past `n` hold +0.0, which a non-negative sum absorbs unchanged, so
// the loop runs in unrolled groups of kWalkUnroll without a remainder.
constexpr int kWalkUnroll = 4;
struct WalkLds {
    double fh[4][64][2];  // [wave][lane]{cfd x count, hsu}
    double j[4][64];      // [wave][lane] jost x count
};
__device__ __forceinline__ void walk_park(WalkLds &w, uint32_t wave, uint32_t lane, double fz, double hz) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the previous chunk's reads are done
    __builtin_amdgcn_wave_barrier();
    w.fh[wave][lane][0] = fz;
    w.fh[wave][lane][1] = hz;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void walk_fold(const WalkLds &w, uint32_t wave, uint32_t n, double &cfd_sum, double &hsu_sum) {
    for (uint32_t l = 0; l < n; l += kWalkUnroll) {
#pragma unroll
        for (int k = 0; k < kWalkUnroll; ++k) {
            cfd_sum += w.fh[wave][l + k][0];
            hsu_sum += w.fh[wave][l + k][1];
        }
    }
}
__device__ __forceinline__ void walk_fold_jost(WalkLds &w, uint32_t wave, uint32_t lane, uint32_t n, double jz, double &jost_sum) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    w.j[wave][lane] = jz;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t l = 0; l < n; l += kWalkUnroll) {
#pragma unroll
        for (int k = 0; k < kWalkUnroll; ++k) jost_sum += w.j[wave][l + k];
    }
}

// One wave per guide.  Integer aggregates are wave reductions (exact in any order); the two f64 sums are accumulated
// hit by hit IN DATABASE ORDER (lane values broadcast one after the other) so that they associate exactly like the
// reference's sequential folds (Doench2016CFDScore.scala:79, CrisprMitEduOffTarget.scala:104).
__global__ __launch_bounds__(256) void k_guide_aggregate(const uint64_t *__restrict__ ret_off, const uint32_t *__restrict__ n_ret,
                                                         const uint32_t *__restrict__ ot_count, const uint32_t *__restrict__ full,
                                                         const uint8_t *__restrict__ mm, const uint32_t *__restrict__ cnt, const double *__restrict__ cfd,
                                                         const double *__restrict__ hsu, const double *__restrict__ jost /* may be null */,
                                                         uint32_t n_guides, GuideSummary *__restrict__ out) {
    __shared__ WalkLds wk;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = blockIdx.x * 4 + wave;
    if (g >= n_guides) return;
    const uint32_t n = n_ret[g];
    const uint64_t b = ret_off[g];
    uint32_t hist[5] = {0, 0, 0, 0, 0}, closest = 0xFFFFFFFFu, n_scored = 0;
    double cfd_sum = 0.0, hsu_sum = 0.0, cfd_max = 0.0, jost_sum = 0.0, jost_max = 0.0, lane_cfd_max = 0.0, lane_jost_max = 0.0;
    for (uint32_t i = 0; i < n; i += 64) {  // pass 1: histogram, closest level, ordered f64 sums
        const bool in = i + lane < n;
        const uint32_t m = in ? mm[b + i + lane] : 0xFFu, c = in ? cnt[b + i + lane] : 0u;
        const double f = in ? cfd[b + i + lane] : __builtin_nan(""), h = in ? hsu[b + i + lane] : 0.0;
#pragma unroll
        for (int k = 0; k < 5; ++k) hist[k] += (m == (uint32_t)k) ? c : 0u;     // ClosestHit.scala:57-59
        if (in && m > 0 && m < closest) closest = m;                               // :62-64
        // the ordered walk of k_guide_epilogue: lanes 0 .. nin-1 in order, +0.0 for the unscored (NaN = the on-target itself)
        const uint32_t nin = min(n - i, 64u);
        const bool sc = in && f == f;
        const double fz = sc ? f * (double)c : 0.0, hz = sc ? h : 0.0;
        lane_cfd_max = fmax(lane_cfd_max, sc ? f : 0.0);                            // scores are >= 0, the empty max is 0.0
        n_scored += (uint32_t)__popcll(__ballot(sc));
        walk_park(wk, wave, lane, fz, hz);
        walk_fold(wk, wave, nin, cfd_sum, hsu_sum);
        if (jost) {                                                                // JostAndSantosCRISPRi.scala:42-43, same walk
            const double j = in ? jost[b + i + lane] : __builtin_nan("");
            const bool sj = in && j == j;
            const double jz = sj ? j * (double)c : 0.0;
            lane_jost_max = fmax(lane_jost_max, sj ? j : 0.0);
            walk_fold_jost(wk, wave, lane, nin, jz, jost_sum);
        }
    }
    cfd_max = wave_max_f64(lane_cfd_max);
    jost_max = wave_max_f64(lane_jost_max);
    closest = wave_min_u32(closest);
    uint32_t closest_count = 0, in_genome = 0;
    for (uint32_t i = 0; i < n; i += 64) {  // pass 2: occurrences at the closest level (ClosestHit.scala:62-67)
        const bool in = i + lane < n;
        const uint32_t m = in ? mm[b + i + lane] : 0xFFu, c = in ? cnt[b + i + lane] : 0u;
        closest_count += (m == closest) ? c : 0u;
    }
    GuideSummary s;
    s.n_hits = n; s.ot_count = ot_count[g]; s.overflow = full[g];
#pragma unroll
    for (int k = 0; k < 5; ++k) s.hist[k] = wave_sum_u32(hist[k]);
    in_genome = s.hist[0];                                                          // DangerousSequences.scala:62
    s.closest = closest;
    s.closest_count = closest == 0xFFFFFFFFu ? 0u : wave_sum_u32(closest_count);
    s.in_genome = in_genome; s.n_scored = n_scored;
    s.cfd_max = cfd_max; s.cfd_sum = cfd_sum; s.hsu_sum = hsu_sum;
    s.jost_max = jost_max; s.jost_sum = jost_sum;
    if (lane == 0) out[g] = s;
}

// The epilogue of a discover call that only wants the per-guide aggregates (FFH_FINALIZE_SUMMARIES_ONLY), one wave per
// guide and one pass over its hits: ordered cut-off (k_cutoff), per-hit scores (k_score_hits) and the aggregation
// (k_guide_aggregate) without any per-hit array in between.  Same arithmetic in the same order, so the summaries are
// bit-identical to the three-kernel path that also delivers the hit lists.
// st == nullptr: the target longs of the hits have not been gathered (k_hit_targets); the kernel then reads them through the
// sorted hit keys itself -- its waves are busy with the ordered walk, so the gather hides behind them instead of costing a pass.
__global__ __launch_bounds__(256) void k_guide_epilogue(const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_end, const uint64_t *__restrict__ st,
                                                        const uint64_t *__restrict__ hit_keys, const uint64_t *__restrict__ targets, int tbits,
                                                        const uint32_t *__restrict__ prior, const uint64_t *__restrict__ guides, Geometry geo,
                                                        const ScoreTables *__restrict__ tab, uint32_t n_guides, uint32_t overflow, int want_jost,
                                                        uint32_t *__restrict__ n_ret, GuideSummary *__restrict__ out,
                                                        uint32_t *__restrict__ totals_out /* nullable: min(positions of all hits, overflow) */,
                                                        const uint32_t *__restrict__ fix_totals /* nullable: redo only guides the prior changes */) {
    __shared__ ScoreTables lt;  // 4.6 KB: the coefficient tables, read with rolled loops (low register count -> 8 waves per SIMD)
    {
        const double *src = reinterpret_cast<const double *>(tab);
        double *dst = reinterpret_cast<double *>(&lt);
        for (uint32_t i = threadIdx.x; i < sizeof(ScoreTables) / sizeof(double); i += blockDim.x) dst[i] = src[i];
    }
    __shared__ WalkLds wk;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = blockIdx.x * 4 + wave;
    if (g >= n_guides) return;
    const uint32_t b = seg_begin[g], e = seg_end[g], p0 = prior ? prior[g] : 0u;
    // multi-GPU fix-up pass: a shard's own aggregates (computed with prior 0) stand unless the positions of the shards before it
    // push this guide's running total to the limit inside or before this shard
    if (fix_totals && !(p0 > 0u && p0 + fix_totals[g] >= overflow)) return;
    const uint64_t gd = guides[g];
    uint32_t run = p0, kept = 0;
    uint32_t hist[5] = {0, 0, 0, 0, 0}, closest = 0xFFFFFFFFu, closest_count = 0, n_scored = 0;
    double cfd_sum = 0.0, hsu_sum = 0.0, cfd_max = 0.0, jost_sum = 0.0, jost_max = 0.0, lane_cfd_max = 0.0, lane_jost_max = 0.0;
    for (uint32_t i = b; i < e && run < overflow; i += 64) {
        const bool in = i + lane < e;
        const uint64_t t = !in ? 0ull : st ? st[i + lane] : targets[hit_keys[i + lane] & ((1ull << tbits) - 1ull)];
        const uint32_t c = in ? (uint32_t)(t >> 48) : 0u;
        const uint32_t incl = wave_inclusive_scan_u32(c, lane);
        const bool keep = in && (run + (incl - c) < overflow);          // CRISPRSiteOT.addOT / full, crispr/CRISPRSiteOT.scala:39-46
        const uint32_t nk = (uint32_t)__popcll(__ballot(keep));
        kept += nk;
        if (nk) run += __shfl(incl, nk - 1, 64);
        int mmi = 0xFF;
        double f = __builtin_nan(""), h = 0.0, j = __builtin_nan("");
        if (keep) {
            score_pair(gd, t, geo, &lt, mmi, f, h);
            if (want_jost && geo.c0 == 3 && mmi != 0) j = jost_pair(gd, t, geo, &lt);
        }
        const uint32_t m = keep ? (uint32_t)mmi : 0xFFu, ck = keep ? c : 0u;
#pragma unroll
        for (int k = 0; k < 5; ++k) hist[k] += (m == (uint32_t)k) ? ck : 0u;  // ClosestHit.scala:57-59
        const uint32_t cm = wave_min_u32((keep && m > 0) ? m : 0xFFFFFFFFu);  // :62-67, folded chunk by chunk
        if (cm < closest) { closest = cm; closest_count = 0; }
        if (cm != 0xFFFFFFFFu && cm == closest) closest_count += wave_sum_u32((m == closest) ? ck : 0u);
        // ordered f64 sums: the kept hits are lanes 0 .. nk-1, walked in that order.  Unscored hits (the on-target itself) add +0.0,
        // which leaves a non-negative sum bit for bit as it is, so the walk needs no mask; maxima do not depend on the order and
        // are kept per lane (one wave reduction at the end).
        const bool sc = keep && f == f;
        const double fz = sc ? f * (double)c : 0.0, hz = sc ? h : 0.0;
        lane_cfd_max = fmax(lane_cfd_max, sc ? f : 0.0);
        n_scored += (uint32_t)__popcll(__ballot(sc));
        walk_park(wk, wave, lane, fz, hz);
        walk_fold(wk, wave, nk, cfd_sum, hsu_sum);
        if (want_jost) {
            const bool sj = keep && j == j;
            const double jz = sj ? j * (double)c : 0.0;
            lane_jost_max = fmax(lane_jost_max, sj ? j : 0.0);
            walk_fold_jost(wk, wave, lane, nk, jz, jost_sum);
        }
    }
    cfd_max = wave_max_f64(lane_cfd_max);
    jost_max = wave_max_f64(lane_jost_max);
    GuideSummary s;
    s.n_hits = kept; s.ot_count = run - p0; s.overflow = run >= overflow;
#pragma unroll
    for (int k = 0; k < 5; ++k) s.hist[k] = wave_sum_u32(hist[k]);
    s.closest = closest;
    s.closest_count = closest == 0xFFFFFFFFu ? 0u : closest_count;
    s.in_genome = s.hist[0]; s.n_scored = n_scored;
    s.cfd_max = cfd_max; s.cfd_sum = cfd_sum; s.hsu_sum = hsu_sum;
    s.jost_max = jost_max; s.jost_sum = jost_sum;
    if (lane == 0) {
        out[g] = s; n_ret[g] = kept;
        if (totals_out) totals_out[g] = min(run - p0, overflow);
    }
}

__global__ void k_gather_positions(const uint32_t *__restrict__ tidx, const uint32_t *__restrict__ cnt, const uint64_t *__restrict__ out_off,
                                   uint64_t n_hits, const uint64_t *__restrict__ db_pos_off, const uint64_t *__restrict__ db_pos,
                                   uint64_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    const uint64_t src = db_pos_off[tidx[i]], dst = out_off[i];
    const uint32_t c = cnt[i];
    for (uint32_t k = 0; k < c; ++k) out[dst + k] = db_pos[src + k];
}

}  // namespace ffh
