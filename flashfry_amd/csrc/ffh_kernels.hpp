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

// A value every lane of the wave holds equal, said so: the compiler cannot see that threadIdx.x >> 6 (or anything loaded through it) is
// wave-uniform, and without this it predicates every loop over a wave's segment per lane
__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

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

// rest key of a target / guide on one side: the planes of the bases its bucket id does NOT hold, H << 16 | L (<= 12 bases each).
// Prefix image (bucket = first a bases): the last lc - a bases; suffix image (bucket = last s bases): the first lc - s bases.
__host__ __device__ __forceinline__ uint32_t prefix_rest_key(uint64_t pk, int lc, int a) {
    const uint32_t m = (1u << (lc - a)) - 1u;
    return ((((uint32_t)(pk >> 32)) & m) << 16) | ((uint32_t)pk & m);
}
__host__ __device__ __forceinline__ uint32_t suffix_rest_key(uint64_t pk, int s) {
    return (((uint32_t)(pk >> 32) >> s) << 16) | ((uint32_t)pk >> s);
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
// bad[0]: counts the reference would refuse; bad[1]: neighbours that are not in sequence order (a database the reference writes is:
// BinWriter / BlockReader sort every bin, the bins are written in prefix order; ffh_db_load_soa takes what it is given)
__global__ void k_check_counts(const uint64_t *__restrict__ targets, uint64_t n, uint64_t seq_mask /* the 2 x scan length sequence bits */,
                               uint32_t *__restrict__ counts, uint32_t *__restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t t = targets[i];
    const uint32_t c = (uint32_t)(t >> 48);
    counts[i] = c;
    if (c == 0 || c > 32767) atomicAdd(bad, 1u);  // getCount is a signed short and must be > 0 (BlockManager.scala:232-234)
    // (bits above the sequence -- some synthetic packs carry them -- would lead a plain comparison and hide an unordered sequence)
    if (i && ((targets[i - 1] & seq_mask) > (t & seq_mask) || ((targets[i - 1] ^ t) & 0xFFFFFFFFFFFFull & ~seq_mask))) atomicAdd(bad + 1, 1u);
}

template <bool SUFFIX>
__global__ void k_image_hist(const uint64_t *__restrict__ targets, uint64_t n, Geometry geo, int width, uint32_t *__restrict__ bcount) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const uint64_t pk = planar_key(live ? targets[i] : 0, geo.c0, geo.lc);
    const uint32_t b = SUFFIX ? suffix_bucket(pk, width) : prefix_bucket(pk, geo.lc, width);
    if (SUFFIX) { if (live) atomicAdd(&bcount[b], 1u); return; }
    // prefix buckets of a database in sequence order come in runs: one atomic per run of a wave, by the run's first lane (one per lane
    // left 3e8 atomics queueing on a few addresses at a time: 20 ms at hg38 scale)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t prev = (uint32_t)__shfl_up((int)b, 1);
    const bool head = live && (lane == 0 || prev != b);
    const uint64_t heads = __ballot(head), alive = __ballot(live);
    if (head) {
        const uint64_t later = lane == 63 ? 0ull : heads >> (lane + 1);                       // the next run's first lane, or the end of the live lanes
        const uint32_t end = later ? lane + 1u + (uint32_t)__builtin_ctzll(later) : (uint32_t)__popcll(alive);
        atomicAdd(&bcount[b], end - lane);
    }
}

template <bool SUFFIX>
__global__ void k_image_scatter(const uint64_t *__restrict__ targets, uint64_t n, Geometry geo, int width,
                                const uint32_t *__restrict__ bstart, uint32_t *__restrict__ bfill, uint32_t *__restrict__ keys,
                                uint32_t *__restrict__ tidx, uint32_t base /* database index of targets[0]: an image over a slab of the database */) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t pk = planar_key(targets[i], geo.c0, geo.lc);
    const uint32_t b = SUFFIX ? suffix_bucket(pk, width) : prefix_bucket(pk, geo.lc, width);
    const uint32_t pos = bstart[b] + atomicAdd(&bfill[b], 1u);
    keys[pos] = SUFFIX ? suffix_rest_key(pk, width) : prefix_rest_key(pk, geo.lc, width);   // the bucket holds the other bases
    tidx[pos] = base + (uint32_t)i;
}

// DIRECT prefix image.  In a database in sequence order whose compared bases lead the sequence (every 3'-PAM pack: the PAM sits in the
// low bits) the targets of a prefix bucket are CONSECUTIVE in the database.  The image then keeps them in database order inside the
// bucket, and the database index of a slot is slot + ddelta[bucket] -- arithmetic on a 4^a-entry table the compare kernel fetches
// with the bucket boundaries -- instead of a 4-byte-per-slot array gathered once per hit (a 128-byte line each: 60 % of the hits of a
// <= 4-mismatch scan come from this image).  first[b] = database index of bucket b's first target.
__global__ void k_bucket_first(const uint64_t *__restrict__ targets, uint64_t n, Geometry geo, int width, uint32_t *__restrict__ first) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = prefix_bucket(planar_key(targets[i], geo.c0, geo.lc), geo.lc, width);
    if (i == 0 || prefix_bucket(planar_key(targets[i - 1], geo.c0, geo.lc), geo.lc, width) != b) first[b] = (uint32_t)i;
}
// one wave per bucket: bit-slices the bucket's targets straight from the database (no counting sort in between) and leaves
// ddelta[b] = first[b] - 32 * gstart[b], so that slot s of the image is target s + ddelta[b]
__global__ __launch_bounds__(256) void k_group_build_direct(const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ first,
                                                            const uint64_t *__restrict__ targets, Geometry geo, int width, uint32_t nb, uint32_t R, uint32_t GW,
                                                            uint32_t *__restrict__ gwords, uint32_t *__restrict__ ddelta) {
    const uint32_t lane = threadIdx.x & 63, b = blockIdx.x * 4 + wave_uniform(threadIdx.x >> 6);
    if (b >= nb) return;
    const uint32_t nt = bstart[b + 1] - bstart[b], g0 = gstart[b], ngr = gstart[b + 1] - g0;
    const uint32_t k0 = nt ? first[b] : 0u;
    if (lane == 0) ddelta[b] = k0 - 32u * g0;
    for (uint32_t c = 0; 2 * c < ngr; ++c) {
        const uint32_t k = c * 64 + lane;
        const bool valid = k < nt;
        const uint32_t key = valid ? prefix_rest_key(planar_key(targets[k0 + k], geo.c0, geo.lc), geo.lc, width) : 0u;
        uint64_t mine = 0;
        for (uint32_t i = 0; i < R; ++i) {
            const uint64_t h = __ballot((key >> (16 + i)) & 1u), l = __ballot((key >> i) & 1u);
            if (lane == 2 * i) mine = h;
            if (lane == 2 * i + 1) mine = l;
        }
        const uint64_t v = __ballot(valid);
        if (lane == 2 * R) mine = v;
        if (lane < GW) {
            gwords[(size_t)(g0 + 2 * c) * GW + lane] = (uint32_t)mine;
            if (2 * c + 1 < ngr) gwords[(size_t)(g0 + 2 * c + 1) * GW + lane] = (uint32_t)(mine >> 32);
        }
    }
}

// The image the compare kernel reads: the targets of a bucket in GROUPS of 32, bit-sliced (ffh_compare.hpp).  A group is GW words:
// word 2i = the high plane bit of rest base i of its 32 targets, word 2i + 1 = the low plane bit, word 2R = which of the 32
// slots hold a target; the database index of slot s of group g is tidx[32 g + s].
__global__ void k_group_count(const uint32_t *__restrict__ bstart, uint32_t nb, uint32_t *__restrict__ gcount) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb) gcount[b] = (bstart[b + 1] - bstart[b] + 31u) >> 5;
}
// one wave per bucket; 64 targets (two groups) per step, transposed with ballots
__global__ __launch_bounds__(256) void k_group_build(const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ keys,
                                                     const uint32_t *__restrict__ tidx_in, uint32_t nb, uint32_t R, uint32_t GW, uint32_t *__restrict__ gwords,
                                                     uint32_t *__restrict__ tidx_out) {
    const uint32_t lane = threadIdx.x & 63, b = blockIdx.x * 4 + wave_uniform(threadIdx.x >> 6);
    if (b >= nb) return;
    const uint32_t k0 = bstart[b], nt = bstart[b + 1] - k0, g0 = gstart[b], ngr = gstart[b + 1] - g0;
    for (uint32_t c = 0; 2 * c < ngr; ++c) {
        const uint32_t k = c * 64 + lane;
        const bool valid = k < nt;
        const uint32_t key = valid ? keys[k0 + k] : 0u;
        if (2 * c + (lane >> 5) < ngr) tidx_out[(size_t)(g0 + 2 * c) * 32 + lane] = valid ? tidx_in[k0 + k] : 0xFFFFFFFFu;
        uint64_t mine = 0;   // lane w collects word w of the two groups (low / high half of the ballot)
        for (uint32_t i = 0; i < R; ++i) {
            const uint64_t h = __ballot((key >> (16 + i)) & 1u), l = __ballot((key >> i) & 1u);
            if (lane == 2 * i) mine = h;
            if (lane == 2 * i + 1) mine = l;
        }
        const uint64_t v = __ballot(valid);
        if (lane == 2 * R) mine = v;
        if (lane < GW) {
            gwords[(size_t)(g0 + 2 * c) * GW + lane] = (uint32_t)mine;
            if (2 * c + 1 < ngr) gwords[(size_t)(g0 + 2 * c + 1) * GW + lane] = (uint32_t)(mine >> 32);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// candidate lists: every guide visits the buckets inside its Hamming ball (key ^ pattern)
// ---------------------------------------------------------------------------------------------------------
// The kernels that build the candidate lists take the arguments of BOTH images (round 5): blockIdx.y picks the image, blocks beyond the
// image's own grid leave at once.  The two images' lists do not depend on each other and most of these launches sit on the launch floor
// (~4.7 us each behind its predecessor), so the pair costs what one cost: 7 launches per step where there were 13.  (Two streams, tried
// in round 3, did not overlap on this stack.)  A single image -- a slab of a bounded scan, a one-image plan -- is launched with grid.y = 1.
struct GuideKeysArgs {
    const uint64_t *guides; uint32_t n; Geometry geo; int width; uint2 *gtab; uint32_t *gbucket;
    uint32_t *seg_begin /* nullable */, *seg_end, *zero_buf; uint32_t n_zero; uint32_t suffix; uint32_t grid;
    unsigned long long *setup_cursor /* nullable */; int first_batch;   // the compare launch's counters, cleared here (compare_setup_words: a launch of its own until round 5)
};
__device__ void compare_setup_words(unsigned long long *__restrict__ cursor, int first_batch, uint32_t t);   // (ffh_compare.hpp)
__global__ void k_guide_keys(GuideKeysArgs a0, GuideKeysArgs a1) {
    const GuideKeysArgs &A = blockIdx.y ? a1 : a0;
    if (blockIdx.x >= A.grid) return;
    if (A.setup_cursor && blockIdx.x == 0 && threadIdx.x < 64) compare_setup_words(A.setup_cursor, A.first_batch, threadIdx.x);
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t d = g; d < A.n_zero; d += A.grid * blockDim.x) A.zero_buf[d] = 0u;  // the partition histogram k_guide_part_hist adds into
    if (g >= A.n) return;
    if (A.seg_begin) { A.seg_begin[g] = 0u; A.seg_end[g] = 0u; }  // the hit segment of a guide without hits (k_segments only visits the others)
    const uint64_t pk = planar_key(A.guides[g], A.geo.c0, A.geo.lc);
    const uint32_t b = A.suffix ? suffix_bucket(pk, A.width) : prefix_bucket(pk, A.geo.lc, A.width);
    A.gbucket[g] = b;
    A.gtab[g] = make_uint2(A.suffix ? suffix_rest_key(pk, A.width) : prefix_rest_key(pk, A.geo.lc, A.width), b);  // what the compare kernel gathers per candidate
}

// Candidate lists in CSR form: for every bucket the ids of the guides whose Hamming ball reaches it (the guides' {rest key,
// bucket} entries stay in a table small enough to live in L2 and are gathered by the compare kernel).  The
// (bucket, guide) entries are enumerated implicitly (bucket = guide bucket ^ pattern) and binned EXACTLY -- no
// capacity guess, so skewed guide sets (tiling libraries, repeats) cost nothing extra -- and without per-entry global
// atomics (device-scope atomics on random addresses run at ~1.3e10/s on MI355X: 4 ms for the 5.6e7 entries of the
// hg38-scale workload):
//   A k_guide_part_hist + the size blocks of k_guide_by_part's launch (k_part_sizes until round 5): the partition (= high bits of the bucket id) sizes, computed without touching
//                           the entries (XOR convolution of two histograms);  exclusive scan of the <= 4096 sizes;
//   B k_guide_by_part + k_item_bin_direct : the guides grouped by partition, then one block per partition enumerates its own
//                           entries from those runs, counts them per bucket in LDS, scans the counts, writes the CSR offsets of
//                           its buckets and puts the guide ids into place (LDS atomics only).
constexpr int kPartThreads = 1024;
constexpr int kMaxPartBits = 12;   // <= 4096 partitions
constexpr int kMaxLowBits = 12;    // <= 4096 buckets per partition (11 bits preferred: see prepare_side)
constexpr int kBinStage = 14336;   // candidate ids staged in LDS per partition (56 KB: two blocks per CU); larger partitions scatter to memory
constexpr int kGidBits = 20;       // guides per batch < 2^20

struct ItemGeom {
    uint32_t n_guides, n_pat;
    uint32_t low_bits;       // bucket id = (partition << low_bits) | low
    uint32_t n_part;
    uint32_t item_base;      // first CSR slot of this image (the two images share the item array)
    const uint32_t *live;    // [2^live_bits] which of the image's bucket-id prefixes of live_bits bits hold a target (k_bucket_live): the
    uint32_t live_bits;      // entries of a partition without one are dropped before they are enumerated.  A bin shard of a multi-GPU
                             // run holds a contiguous eighth of SEQUENCE space; the bucket id keeps the two bit planes apart, so that
                             // is not a range of ids -- but the partition bits are the first bases' high bits and the first base's
                             // low bit, and exactly the partitions of the shard's leading bases are alive: an eighth of the entries
                             // for an eighth of the database (round 3 dropped by {first, last} non-empty bucket: half of them)
    // one slab of a bounded scan (prefix image only): keep the entries whose bucket's first three bases, read as a number 0..63 in
    // sequence order, lie in [rank_lo, rank_hi].  The bucket id holds the planes apart (all high bits, then all low bits), so a
    // slab of the database order is not a range of bucket ids; {0, 63} = everything, and the partition sizes then still come from
    // the size blocks of k_guide_by_part's launch.
    uint32_t rank_lo, rank_hi, width;
};

// the first three bases of a prefix bucket as a number in sequence (= database) order
__device__ __forceinline__ uint32_t bucket_rank(uint32_t b, uint32_t width) {
    const uint32_t h = (b >> (2u * width - 3u)) & 7u, l = (b >> (width - 3u)) & 7u;
    return ((h & 4u) << 3) | ((l & 4u) << 2) | ((h & 2u) << 2) | ((l & 2u) << 1) | ((h & 1u) << 1) | (l & 1u);
}
// which bucket-id prefixes of `bits` bits hold a target (live[] zeroed before)
__global__ void k_bucket_live(const uint32_t *__restrict__ bstart, uint32_t nb, uint32_t shift, uint32_t *__restrict__ live) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb && bstart[b + 1] != bstart[b]) live[b >> shift] = 1u;
}
// does partition `part` (the high part_bits bits of a bucket id) hold a target?  Called by a whole wave (lanes share the flags).
__device__ __forceinline__ bool part_live_wave(const ItemGeom &ig, uint32_t part, uint32_t part_bits, uint32_t lane) {
    const uint32_t sh = ig.live_bits - part_bits, n = 1u << sh;   // (live_bits >= part_bits: prepare_side)
    bool any = false;
    for (uint32_t k = lane; k < n; k += 64) any |= ig.live[(part << sh) + k] != 0u;
    return __ballot(any) != 0ull;
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
struct PartHistArgs { const uint32_t *gbucket; uint32_t n_guides, low_bits, n_part; uint32_t *ghist, *part_fill; uint32_t n_fill, grid; };
__global__ __launch_bounds__(1024) void k_guide_part_hist(PartHistArgs a0, PartHistArgs a1) {
    const PartHistArgs &A = blockIdx.y ? a1 : a0;
    if (blockIdx.x >= A.grid) return;
    __shared__ uint32_t h[1 << kMaxPartBits];
    if (blockIdx.x == 0)
        for (uint32_t d = threadIdx.x; d < A.n_fill; d += blockDim.x) A.part_fill[d] = 0;  // the counters of the passes that follow (saves a fill launch)
    for (uint32_t d = threadIdx.x; d < A.n_part; d += blockDim.x) h[d] = 0;
    __syncthreads();
    const uint32_t per = (A.n_guides + A.grid - 1) / A.grid, g_end = min(A.n_guides, (blockIdx.x + 1) * per);
    for (uint32_t g = blockIdx.x * per + threadIdx.x; g < g_end; g += blockDim.x) atomicAdd(&h[lds_slot(A.gbucket[g] >> A.low_bits)], 1u);
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < A.n_part; d += blockDim.x) {
        const uint32_t c = h[lds_slot(d)];
        if (c) atomicAdd(&A.ghist[d], c);
    }
}
// (the partitions' entry counts: a wave per partition, the lanes striding over the patterns -- a thread per partition left 4096 threads
// walking 529 patterns each: 36 us --, in the blocks behind the sorting ones of k_guide_by_part's launch)

// ---------------------------------------------------------------------------------------------------------
// The CSR is built without intermediate records (round 3; round 2 wrote every entry as a 4-byte record into one of 4096 runs first and
// read it back: 0.33 ms for the prefix side).  bucket = guide bucket ^ pattern acts on the partition bits and on the low
// bits separately, so the entries of partition q come, for every pattern p, from the guides of ONE partition, q ^ high(p).  With the
// guides grouped by partition (k_guide_by_part: a counting sort on the histogram k_guide_part_hist leaves anyway; 100 000 guides)
// the block of partition q enumerates its own entries straight from those runs -- once to count its buckets, once to place the
// guide ids: ~0.15 ms for the 5.3e7 entries of the hg38-scale prefix image.
// ---------------------------------------------------------------------------------------------------------
// guides grouped by partition: by_part[gp_start[part] + k] = (low bucket bits << kGidBits) | guide.  A block of 1024 guides counts
// its guides per partition in LDS and reserves one run per partition it touches (an image with few partitions -- the suffix side --
// put ~400 same-address global atomics on every counter: 47 us; this way a few per block)
// (round 5: the exclusive scan of the partition histogram -- <= 4096 counters -- is done by every block for itself in LDS instead of in
// two launches of its own; block 0 leaves it in gp_start for k_item_bin_direct)
// (... and the partitions' entry counts, which need the same histogram and nothing else, are computed by the blocks from `sort_blocks` on
// of the same launch -- k_part_sizes' work, a wave per partition: one launch less per image and step.  patterns == nullptr: not wanted,
// a slab of a bounded scan counts its entries with k_item_bin_direct<true, true>.)
struct ByPartArgs {
    const uint32_t *gbucket; uint32_t n_guides, low_bits, n_part; const uint32_t *part_hist; uint32_t *gp_start /* [n_part + 1] out */, *gp_fill /* zeroed */, *by_part;
    uint32_t sort_blocks; const uint32_t *patterns; ItemGeom ig; uint32_t part_bits; uint32_t *part_count; uint32_t grid;
};
__global__ __launch_bounds__(1024) void k_guide_by_part(ByPartArgs a0, ByPartArgs a1) {
    const ByPartArgs &A = blockIdx.y ? a1 : a0;
    if (blockIdx.x >= A.grid) return;
    const uint32_t *__restrict__ gbucket = A.gbucket, *__restrict__ part_hist = A.part_hist, *__restrict__ patterns = A.patterns;
    uint32_t *__restrict__ gp_start = A.gp_start, *__restrict__ gp_fill = A.gp_fill, *__restrict__ by_part = A.by_part, *__restrict__ part_count = A.part_count;
    const uint32_t n_guides = A.n_guides, low_bits = A.low_bits, n_part = A.n_part, sort_blocks = A.sort_blocks, part_bits = A.part_bits;
    const ItemGeom &ig = A.ig;
    __shared__ uint32_t cnt[1 << kMaxPartBits];
    __shared__ uint32_t start[1 << kMaxPartBits];
    __shared__ uint32_t scan_lds[16];
    if (blockIdx.x >= sort_blocks) {
        const uint32_t q = (blockIdx.x - sort_blocks) * 16u + wave_uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (q >= ig.n_part) return;
        uint32_t n = 0;
        for (uint32_t p = lane; p < ig.n_pat; p += 64) n += part_hist[q ^ (patterns[p] >> ig.low_bits)];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
        const bool live = part_live_wave(ig, q, part_bits, lane);
        if (lane == 0) part_count[q] = live ? n : 0u;   // partitions without a target take no entries
        return;
    }
    {   // start[q] = guides in the partitions before q: four counters per thread
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t q = 4u * threadIdx.x + (uint32_t)k; c[k] = q < n_part ? part_hist[q] : 0u; sum += c[k]; }
        uint32_t tot;
        uint32_t off = block_exclusive_scan_1024(sum, scan_lds, tot);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t q = 4u * threadIdx.x + (uint32_t)k;
            if (q < n_part) { start[q] = off; cnt[q] = 0; if (blockIdx.x == 0) gp_start[q] = off; }
            off += c[k];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) gp_start[n_part] = tot;
    }
    __syncthreads();
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t b = 0, part = 0, local = 0;
    if (g < n_guides) { b = gbucket[g]; part = b >> low_bits; local = atomicAdd(&cnt[part], 1u); }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < n_part; q += blockDim.x) {
        const uint32_t c = cnt[q];
        if (c) cnt[q] = start[q] + atomicAdd(&gp_fill[q], c);
    }
    __syncthreads();
    if (g < n_guides) by_part[cnt[part] + local] = ((b & ((1u << low_bits) - 1u)) << kGidBits) | g;
}

// One block per partition.  COUNT: only the number of entries the partition keeps (the exact sizes a slab of a bounded scan needs:
// its rank filter looks at the low bucket bits, so the XOR convolution of k_part_sizes does not apply) -> part_count[d].
struct ItemBinArgs {
    const uint32_t *gp_start, *by_part, *patterns; ItemGeom ig; uint32_t part_bits; const uint32_t *part_size /* entries per partition (placing) */;
    uint32_t *part_count, *istart, *item_gid; const uint32_t *bstart; unsigned long long *part_pairs; uint32_t grid;
};
template <bool COUNT, bool SLAB>
__global__ __launch_bounds__(kPartThreads, 8) void k_item_bin_direct(ItemBinArgs a0, ItemBinArgs a1) {
    const ItemBinArgs &A = blockIdx.y ? a1 : a0;
    if (blockIdx.x >= A.grid) return;
    const uint32_t *__restrict__ gp_start = A.gp_start, *__restrict__ by_part = A.by_part, *__restrict__ patterns = A.patterns, *__restrict__ part_size = A.part_size,
                   *__restrict__ bstart = A.bstart;
    uint32_t *__restrict__ part_count = A.part_count, *__restrict__ istart = A.istart, *__restrict__ item_gid = A.item_gid;
    unsigned long long *__restrict__ part_pairs = A.part_pairs;
    const ItemGeom &ig = A.ig;
    const uint32_t part_bits = A.part_bits;
    __shared__ uint32_t cnt[1 << kMaxLowBits];
    __shared__ uint32_t stage[COUNT ? 1 : kBinStage];
    __shared__ uint32_t scan_lds[16];
    __shared__ unsigned long long pair_lds[16];
    const uint32_t d = blockIdx.x, nlow = 1u << ig.low_bits, lowmask = nlow - 1u;
    const bool live = part_live_wave(ig, d, part_bits, threadIdx.x & 63u);   // partitions without a target take no entries
    // virtual threads = (pattern, slice of the source run): S slices per pattern keep the block busy when there are few patterns.
    // (A wave per slice of the patterns with its lanes side by side on the source run -- coalesced reads -- was measured too: 318
    // against 210 us for the hg38-scale prefix image; the runs are short (~24 guides) and a wave then walks ~33 patterns one after
    // the other.)  fn(low bucket bits, guide).
    const uint32_t S = max(1u, (2u * (uint32_t)kPartThreads) / max(ig.n_pat, 1u)), V = ig.n_pat * S;
    auto enumerate = [&](auto &&fn) {
        if (!live) return;
        for (uint32_t v = threadIdx.x; v < V; v += kPartThreads) {
            const uint32_t p = v / S, sl = v - p * S;
            const uint32_t pat = patterns[p], src = d ^ (pat >> ig.low_bits), plo = pat & lowmask;
            const uint32_t k0 = gp_start[src], k1 = gp_start[src + 1];
            // four records of the run per trip, requested together: the kernel is a chain of dependent L2 round trips (pattern -> run
            // bounds -> records) at full occupancy, so what shortens it is loads in flight per thread, not more threads
            for (uint32_t k = k0 + sl; k < k1; k += 4u * S) {
                uint32_t rec[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) rec[u] = k + (uint32_t)u * S < k1 ? by_part[k + (uint32_t)u * S] : 0xFFFFFFFFu;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k + (uint32_t)u * S >= k1) break;
                    const uint32_t low = (rec[u] >> kGidBits) ^ plo;
                    if (SLAB) {
                        const uint32_t r = bucket_rank((d << ig.low_bits) | low, ig.width);
                        if (r < ig.rank_lo || r > ig.rank_hi) continue;
                    }
                    fn(low, rec[u] & ((1u << kGidBits) - 1u));
                }
            }
        }
    };
    for (uint32_t l = threadIdx.x; l < nlow; l += kPartThreads) cnt[l] = 0;
    __syncthreads();
    enumerate([&](uint32_t low, uint32_t) { atomicAdd(&cnt[lds_slot(low)], 1u); });
    __syncthreads();
    const uint32_t per = (nlow + kPartThreads - 1) / kPartThreads, l0 = threadIdx.x * per;
    if (COUNT) {
        uint32_t mine = 0;
        for (uint32_t k = 0; k < per; ++k)
            if (l0 + k < nlow) mine += cnt[lds_slot(l0 + k)];
        uint32_t tot;
        (void)block_exclusive_scan_1024(mine, scan_lds, tot);
        if (threadIdx.x == 0) part_count[d] = tot;
        return;
    }
    // the entries of the partitions before this one: every block adds up its own prefix of the <= 4096 sizes (an exclusive scan in one
    // or two launches of its own until round 5)
    uint32_t p0 = 0;
    {
        uint32_t before = 0;
        for (uint32_t q = threadIdx.x; q < d; q += kPartThreads) before += part_size[q];
        (void)block_exclusive_scan_1024(before, scan_lds, p0);
    }
    const uint32_t n = part_size[d];
    const uint32_t gbase = ig.item_base + p0;  // the entries of partition d occupy CSR slots [item_base + p0, + n)
    {   // exclusive scan of the nlow counters: CSR offsets out, counters become placement cursors; the executed-pair statistic
        uint32_t mine = 0;
        for (uint32_t k = 0; k < per; ++k)
            if (l0 + k < nlow) mine += cnt[lds_slot(l0 + k)];
        uint32_t tot;
        uint32_t off = block_exclusive_scan_1024(mine, scan_lds, tot);
        unsigned long long pairs = 0;
        for (uint32_t k = 0; k < per; ++k)
            if (l0 + k < nlow) {
                const uint32_t c = cnt[lds_slot(l0 + k)];
                const uint64_t bucket = ((uint64_t)d << ig.low_bits) + l0 + k;
                istart[bucket] = gbase + off;
                cnt[lds_slot(l0 + k)] = off;
                off += c;
                if (c) pairs += (unsigned long long)c * (bstart[bucket + 1] - bstart[bucket]);
            }
        if (d == A.grid - 1 && threadIdx.x == kPartThreads - 1) istart[(uint64_t)ig.n_part << ig.low_bits] = gbase + n;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) pairs += __shfl_xor(pairs, sft, 64);
        if ((threadIdx.x & 63) == 0) pair_lds[threadIdx.x >> 6] = pairs;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < kPartThreads / 64; ++w) t += pair_lds[w];
            part_pairs[d] = t;
        }
    }
    __syncthreads();
    if (n <= (uint32_t)kBinStage) {   // the usual case: the ids are put in bucket order inside LDS and leave as one coalesced copy
        enumerate([&](uint32_t low, uint32_t g) { stage[atomicAdd(&cnt[lds_slot(low)], 1u)] = g; });
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += kPartThreads) item_gid[gbase + k] = stage[k];
    } else {
        enumerate([&](uint32_t low, uint32_t g) { item_gid[gbase + atomicAdd(&cnt[lds_slot(low)], 1u)] = g; });
    }
}

// (the compare kernel lives in ffh_compare.hpp)

// ---------------------------------------------------------------------------------------------------------
// epilogue: ordered cut-off (crispr/CRISPRSiteOT.scala:39-46) + per-hit scores + per-guide aggregates.
// The hits are sorted by (guide, database index); one WAVE works on one guide's segment.
// ---------------------------------------------------------------------------------------------------------
// (the hit buffer ends with the all-ones padding of the compare waves' last chunks: guide field >= n_guides, skipped everywhere)
__global__ void k_segments(const uint64_t *__restrict__ hits, uint64_t n, int tbits, uint32_t n_guides, uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t g = hits[i] >> tbits;
    if (g >= n_guides) return;
    if (i == 0 || (hits[i - 1] >> tbits) != g) seg_begin[g] = (uint32_t)i;
    if (i == n - 1 || (hits[i + 1] >> tbits) != g) seg_end[g] = (uint32_t)(i + 1);
}

// the target long of every raw hit, in sorted order (the one random gather of the epilogue)
__global__ void k_hit_targets(const uint64_t *__restrict__ hits, uint64_t n, int tbits, uint32_t n_guides, const uint64_t *__restrict__ targets, uint64_t *__restrict__ st) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = hits[i];
    st[i] = (key >> tbits) < n_guides ? targets[key & ((1ull << tbits) - 1ull)] : 0ull;
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
                                                uint32_t *__restrict__ ot_count, uint32_t *__restrict__ full, uint32_t *__restrict__ totals,
                                                uint32_t *__restrict__ pre /* nullable: per kept hit, the guide's kept positions before it */) {
    const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * 4 + wave_uniform(threadIdx.x >> 6);
    if (g >= n_guides) return;
    const uint32_t b = wave_uniform(seg_begin[g]), e = wave_uniform(seg_end[g]), p0 = wave_uniform(prior ? prior[g] : 0u);
    uint32_t run = p0, kept = 0;
    for (uint32_t i = b; i < e && run < overflow; i += 64) {
        const bool in = i + lane < e;
        const uint32_t c = in ? (uint32_t)(st[i + lane] >> 48) : 0u;
        const uint32_t incl = wave_inclusive_scan_u32(c, lane);
        const bool keep = in && (run + (incl - c) < overflow);
        if (pre && keep) pre[i + lane] = run - p0 + (incl - c);
        const uint32_t nk = (uint32_t)__popcll(__ballot(keep));
        kept += nk;
        if (nk) run += (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(nk - 1u));   // (nk is a ballot's popcount: uniform)
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
__global__ void k_score_hits(const uint64_t *__restrict__ hits, uint64_t n_hits, int tbits, uint32_t n_guides, const uint32_t *__restrict__ seg_begin,
                             const uint32_t *__restrict__ n_ret, const uint64_t *__restrict__ ret_off, const uint64_t *__restrict__ st,
                             const uint64_t *__restrict__ guides, Geometry geo, const ScoreTables *__restrict__ tab, uint64_t *__restrict__ out_target,
                             uint8_t *__restrict__ out_mm, uint32_t *__restrict__ out_cnt, uint32_t *__restrict__ out_tidx,
                             double *__restrict__ out_cfd, double *__restrict__ out_hsu, double *__restrict__ out_jost /* may be null */,
                             const uint32_t *__restrict__ pre, const uint64_t *__restrict__ pos_base, uint64_t *__restrict__ out_posoff /* all three
                             nullable: first position slot of every retained hit = the guide's base + the kept positions before the hit (k_cutoff) */,
                             uint64_t *__restrict__ h_target = nullptr, uint8_t *__restrict__ h_mm = nullptr, double *__restrict__ h_cfd = nullptr /* nullable
                             (FFH_LIST_ZERO_COPY): the result's own page-locked arrays -- what the host reads is stored there as well, under the kernel */) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    const uint64_t key = hits[i];
    if ((key >> tbits) >= n_guides) return;   // chunk padding
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
    if (out_posoff) out_posoff[o] = pos_base[g] + pre[i];
    if (h_target) h_target[o] = t;
    if (h_mm) h_mm[o] = (uint8_t)mm;
    if (h_cfd) h_cfd[o] = cfd;
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
    const uint32_t lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6), g = blockIdx.x * 4 + wave;
    if (g >= n_guides) return;
    const uint32_t n = wave_uniform(n_ret[g]);
    const uint64_t b = ((uint64_t)wave_uniform((uint32_t)(ret_off[g] >> 32)) << 32) | wave_uniform((uint32_t)ret_off[g]);
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
// (Round 3 tried persistent blocks -- tables filled once per block, every wave looping over guides with the bounds / keys / target
// longs of the next three guides requested ahead: 95 registers instead of 64, five waves per SIMD instead of eight, and 0.345
// against 0.31 ms at hg38 scale, 0.34 against 0.29 ms on an eighth of it.  The chain of dependent loads is hidden better by the
// three extra waves than by the prefetch.  Two or four guides per wave one after the other, the tables filled once per 8 or 16 guides:
// 0.335 against 0.313 ms.)
__global__ __launch_bounds__(256) void k_guide_epilogue(const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_end, const uint64_t *__restrict__ st,
                                                        const uint64_t *__restrict__ hit_keys, const uint64_t *__restrict__ targets, int tbits,
                                                        const uint32_t *__restrict__ prior, const uint64_t *__restrict__ guides, Geometry geo,
                                                        const ScoreTables *__restrict__ tab, uint32_t n_guides, uint32_t overflow, int want_jost,
                                                        uint32_t *__restrict__ n_ret, GuideSummary *__restrict__ out,
                                                        uint32_t *__restrict__ totals_out /* nullable: min(positions of all hits, overflow) */,
                                                        const uint32_t *__restrict__ fix_totals /* nullable: redo only guides the prior changes */,
                                                        GuideSummary *__restrict__ host_out /* nullable: page-locked host copy of out[], written
                                                        by the kernel itself so that no device-to-host copy follows the launch */) {
    __shared__ ScoreTables lt;  // 4.6 KB: the coefficient tables, read with rolled loops (low register count -> 8 waves per SIMD)
    {
        const double *src = reinterpret_cast<const double *>(tab);
        double *dst = reinterpret_cast<double *>(&lt);
        for (uint32_t i = threadIdx.x; i < sizeof(ScoreTables) / sizeof(double); i += blockDim.x) dst[i] = src[i];
    }
    __shared__ WalkLds wk;
    __shared__ GuideSummary out_lds[4];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6), g = blockIdx.x * 4 + wave;
    if (g >= n_guides) return;
    const uint32_t b = wave_uniform(seg_begin[g]), e = wave_uniform(seg_end[g]), p0 = wave_uniform(prior ? prior[g] : 0u);
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
        if (nk) run += (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(nk - 1u));   // (nk is a ballot's popcount: uniform)
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
    // The 88 bytes leave as ONE coalesced store of 22 lanes (through the wave's LDS slot) instead of six 16-byte stores of lane 0: the
    // copy in page-locked host memory crosses PCIe, where a 16-byte write costs a packet of its own -- 100 000 summaries took the
    // kernel ~0.3 ms whatever the number of hits (a shard with an eighth of them: 0.306 against 0.308 ms).
    if (lane == 0) {
        out_lds[wave] = s; n_ret[g] = kept;
        if (totals_out) totals_out[g] = min(run - p0, overflow);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    static_assert(sizeof(GuideSummary) == 88, "22 words");
    if (lane < 22) {
        const uint32_t v = reinterpret_cast<const uint32_t *>(&out_lds[wave])[lane];
        reinterpret_cast<uint32_t *>(out + g)[lane] = v;
        if (host_out) reinterpret_cast<uint32_t *>(host_out + g)[lane] = v;
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

// ---------------------------------------------------------------------------------------------------------
// Bounded scan (ffh_scan_bounded): the database is scanned slab by slab in database order; after every slab a guide whose
// positions so far reach maximumOffTargets is retired -- the reference stops feeding such a guide as well
// (crispr/ResultsAggregator.scala:61-69, LinearTraversal.scala:64-76).
// ---------------------------------------------------------------------------------------------------------
// total[g] += positions of the slab just scanned (both saturated at the limit); flag[g] = still below the limit
// gtab0 (nullable): the prefix image's guide table when its candidate list is shared by all slabs -- a guide that retires now gets
// the COMPLEMENT of its bucket id there: against every bucket of its candidate list (its own id with <= r1 bases changed) the key
// then differs in >= width - r1 bases, more than any maxMismatch a two-image plan runs with, so the compare kernel gives its jobs
// no steps -- without a test of its own in the hot loop.
// (slab_total64: the slab's totals as k_slab_totals leaves them, unsaturated sums; else slab_total, k_cutoff's)
// allow (nullable): for a guide that reaches the limit IN this slab, the positions it still had to go when the slab began (0: any other)
// (slab_total64[g] is left ZERO for the next slab's k_slab_totals: one memset per scan instead of one per slab)
__global__ void k_bound_update(uint32_t *__restrict__ total, const uint32_t *__restrict__ slab_total, unsigned long long *__restrict__ slab_total64, uint32_t n,
                               uint32_t limit, uint32_t *__restrict__ flag, uint2 *__restrict__ gtab0, uint32_t key_mask, uint32_t *__restrict__ allow) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t slab = slab_total64 ? (uint32_t)min(slab_total64[g], (unsigned long long)limit) : min(limit, slab_total[g]);
    if (slab_total64) slab_total64[g] = 0ull;
    const uint32_t before = total[g], t = min(limit, before + slab);
    if (allow) allow[g] = before < limit && t >= limit ? limit - before : 0u;
    total[g] = t;
    flag[g] = t < limit ? 1u : 0u;
    if (gtab0 && before < limit && t >= limit) gtab0[g].y = ~gtab0[g].y & key_mask;
}
// The positions a slab's raw hits add to every guide, from the records AS THE COMPARE LAUNCH LEFT THEM (any order, chunk padding
// included): the hit's target long is gathered for its count (the same random line k_hit_targets fetched) and the counts are added up per
// guide in an LDS table of the block -- a repeat family's hits arrive together, work entry by work entry, so most of a block's 16 384
// records share a few hundred guides -- and the table's sums leave with one 64-bit atomic per guide and block.  A record that finds no
// slot within eight probes adds its count directly.  (Until round 5 the slab's records were ordered by guide first -- two passes of the
// device-wide sort, k_segments, k_hit_targets, k_cutoff: 0.85 of the repeat-structured workload's 8.05 ms, profiles/r05/ab_log.txt 14.)
constexpr uint32_t kTotThreads = 1024, kTotRows = 16, kTotSlots = 8192, kTotProbes = 8, kTotChunk = kTotThreads * kTotRows;
// sums per 32-bit key in the LDS of a block (one chunk of 16 384 records), flushed to 64-bit global counters
// (Tried on the repeat-structured workload, none of it measurable: blocks that keep their table over a run of chunks; the equal keys of
// neighbouring lanes added up before the table is touched.  The sums of the second pass below are dominated by keys that occur once
// or twice per chunk -- a guide's hits from the suffix image arrive in no order of the index -- i.e. by the flush's ~7e6 global atomics.)
template <uint32_t SLOTS>
struct LdsSums {
    static_assert(2 * sizeof(uint32_t) * SLOTS <= kLdsPerBlock, "LdsSums: the table must fit gfx950's LDS");
    uint32_t tag[SLOTS], sum[SLOTS];
    __device__ __forceinline__ void clear() {
        for (uint32_t i = threadIdx.x; i < SLOTS; i += blockDim.x) { tag[i] = 0xFFFFFFFFu; sum[i] = 0u; }
    }
    __device__ __forceinline__ void add(uint32_t key, uint32_t v, unsigned long long *__restrict__ global) {
        uint32_t s = (key * 2654435761u) >> (32 - __builtin_ctz(SLOTS));
        for (uint32_t k = 0; k < kTotProbes; ++k, s = (s + 1u) & (SLOTS - 1u)) {
            const uint32_t old = atomicCAS(&tag[s], 0xFFFFFFFFu, key);
            if (old == 0xFFFFFFFFu || old == key) { atomicAdd(&sum[s], v); return; }
        }
        atomicAdd(&global[key], (unsigned long long)v);   // (no slot within kTotProbes)
    }
    __device__ __forceinline__ void flush(unsigned long long *__restrict__ global) {
        for (uint32_t i = threadIdx.x; i < SLOTS; i += blockDim.x)
            if (tag[i] != 0xFFFFFFFFu && sum[i]) atomicAdd(&global[tag[i]], (unsigned long long)sum[i]);
    }
};
__global__ __launch_bounds__(kTotThreads) void k_slab_totals(const uint64_t *__restrict__ hits, uint64_t n, int tbits, uint32_t n_guides, const uint64_t *__restrict__ targets,
                                                             unsigned long long *__restrict__ totals /* zeroed */, uint16_t *__restrict__ cnt_out /* nullable; per record: its position count */) {
    __shared__ LdsSums<kTotSlots> T;
    T.clear();
    const uint64_t base = (uint64_t)blockIdx.x * kTotChunk, mask = (1ull << tbits) - 1ull;
    uint64_t key[kTotRows];
    uint32_t cnt[kTotRows];
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r) {
        const uint64_t i = base + (uint64_t)r * kTotThreads + threadIdx.x;
        key[r] = i < n ? hits[i] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r) cnt[r] = (key[r] >> tbits) < n_guides ? (uint32_t)(targets[key[r] & mask] >> 48) : 0u;   // (16 gathers in flight per lane)
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r) {
        const uint64_t i = base + (uint64_t)r * kTotThreads + threadIdx.x;
        if (cnt_out && i < n) cnt_out[i] = (uint16_t)cnt[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r)
        if ((key[r] >> tbits) < n_guides) T.add((uint32_t)(key[r] >> tbits), cnt[r], totals);
    __syncthreads();
    T.flush(totals);
}
// ---- stopping a guide INSIDE the slab in which it reaches the limit ----
// The ordered cut-off keeps a guide's hits, in database order, while the positions before a hit are below the limit; what lies behind
// the hit that reaches it is never delivered.  Retiring guides slab by slab leaves all of the guide's hits of that last slab in the
// records -- 2.7 raw hits per kept one on the repeat-structured workload, all of them ordered by the five-pass sort.  So, for the guides
// that reach the limit in the slab just scanned (k_bound_update: allow[g] > 0), the slab's index span is cut into kSubRanges equal parts,
// the guide's positions are added up per part (k_slab_subhist: the counts k_slab_totals left per record, an LDS table per block like
// there), thr[g] = the part in which the running total reaches what the guide had left (k_slab_threshold), and the slab's records are
// copied out without the chunk padding and without the records of parts behind thr[g] (k_slab_filter, k_slab_keep).  Every dropped
// record has, before it in database order, records of its guide that reach the limit: the delivered lists do not change.
constexpr uint32_t kSubRanges = 32, kSubSlots = 8192;
// the part of the slab an index lies in: any non-decreasing function of the index does, as long as every kernel uses the same one --
// a multiplication by scale = 2^32 kSubRanges / (slab length), worked out on the host (sub_scale), instead of a 64-bit division per record
__device__ __forceinline__ uint32_t sub_range(uint32_t idx, uint32_t lo, uint32_t scale) {
    return min(kSubRanges - 1u, (uint32_t)(((uint64_t)(idx - lo) * scale) >> 32));
}
inline uint32_t sub_scale(uint64_t slab_len) { return (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, ((uint64_t)kSubRanges << 32) / std::max<uint64_t>(slab_len, 1)); }
__global__ __launch_bounds__(kTotThreads) void k_slab_subhist(const uint64_t *__restrict__ hits, const uint16_t *__restrict__ cnt, uint64_t n, int tbits, uint32_t n_guides,
                                                              const uint32_t *__restrict__ allow, uint32_t slab_lo, uint32_t slab_scale,
                                                              unsigned long long *__restrict__ hist /* [n_guides][kSubRanges], zeroed */) {
    __shared__ LdsSums<kSubSlots> T;
    T.clear();
    const uint64_t base = (uint64_t)blockIdx.x * kTotChunk, mask = (1ull << tbits) - 1ull;
    uint64_t key[kTotRows];
    uint32_t a[kTotRows], v[kTotRows];
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r) {
        const uint64_t i = base + (uint64_t)r * kTotThreads + threadIdx.x;
        key[r] = i < n ? hits[i] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r) a[r] = (key[r] >> tbits) < n_guides ? allow[key[r] >> tbits] : 0u;
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r) v[r] = a[r] ? (uint32_t)cnt[base + (uint64_t)r * kTotThreads + threadIdx.x] : 0u;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r)
        if (a[r]) T.add((uint32_t)(key[r] >> tbits) * kSubRanges + sub_range((uint32_t)(key[r] & mask), slab_lo, slab_scale), v[r], hist);
    __syncthreads();
    T.flush(hist);
}
// thr[g] = the last part of the slab whose records guide g keeps (kSubRanges: all of them)
// The rows of hist that k_slab_subhist added to (allow[g] > 0: no other) are left ZERO for the next slab, and so is the filter's counter
// of kept records: one memset of each per scan instead of one per slab (three launches of ~5 us fewer between two compare launches).
__global__ void k_slab_threshold(unsigned long long *__restrict__ hist, const uint32_t *__restrict__ allow, uint32_t n_guides, uint8_t *__restrict__ thr,
                                 unsigned long long *__restrict__ kept) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) *kept = 0ull;
    if (g >= n_guides) return;
    const uint32_t a = allow[g];
    uint32_t t = kSubRanges;
    if (a) {
        unsigned long long run = 0;
        for (uint32_t s = 0; s < kSubRanges; ++s) {
            run += hist[(uint64_t)g * kSubRanges + s];
            hist[(uint64_t)g * kSubRanges + s] = 0ull;
            if (run >= a && t == kSubRanges) t = s;
        }
    }
    thr[g] = (uint8_t)t;
}
// the slab's records [0, n) -> out[0 .. *kept): hits only, and of a guide with a threshold only those of the parts up to it; any order
__global__ __launch_bounds__(kTotThreads) void k_slab_filter(const uint64_t *__restrict__ hits, uint64_t n, int tbits, uint32_t n_guides, const uint8_t *__restrict__ thr,
                                                             uint32_t slab_lo, uint32_t slab_scale, uint64_t *__restrict__ out, unsigned long long *__restrict__ kept /* zeroed */) {
    const uint64_t base = (uint64_t)blockIdx.x * (kTotThreads * kTotRows), mask = (1ull << tbits) - 1ull;
    uint64_t key[kTotRows];
    uint64_t keep_mask[kTotRows];   // (wave-uniform: the row's ballot)
    uint32_t tot = 0;
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r) {
        const uint64_t i = base + (uint64_t)r * kTotThreads + threadIdx.x;
        key[r] = i < n ? hits[i] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r) {
        const uint32_t g = (uint32_t)(key[r] >> tbits);
        bool keep = g < n_guides;
        if (keep) { const uint32_t t = thr[g]; keep = t >= kSubRanges || sub_range((uint32_t)(key[r] & mask), slab_lo, slab_scale) <= t; }
        keep_mask[r] = __ballot(keep);
        tot += (uint32_t)__popcll(keep_mask[r]);
    }
    // ONE reservation per block (same-address atomics complete at ~90 per microsecond: one per wave, 46 000 of them, took 0.5 ms of the
    // repeat-structured step); every row's kept records leave side by side
    __shared__ uint32_t wave_tot[kTotThreads / 64];
    __shared__ unsigned long long block_at;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_tot[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t all = 0;
        for (uint32_t w = 0; w < kTotThreads / 64; ++w) all += wave_tot[w];
        block_at = all ? atomicAdd(kept, (unsigned long long)all) : 0ull;
    }
    __syncthreads();
    unsigned long long at = block_at;
    for (uint32_t w = 0; w < wave; ++w) at += wave_tot[w];
#pragma unroll
    for (int r = 0; r < (int)kTotRows; ++r) {
        const uint64_t m = keep_mask[r];
        if ((m >> lane) & 1ull) out[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key[r];
        at += (uint32_t)__popcll(m);
    }
}
// ... back to where the slab's records began, and the hit cursor / hit count set to what is left (the next compare launch appends there)
__global__ void k_slab_keep(const uint64_t *__restrict__ from, uint64_t *__restrict__ to, const unsigned long long *__restrict__ kept, unsigned long long *__restrict__ counters,
                            unsigned long long cursor_at_start, unsigned long long hits_at_start) {
    const unsigned long long n = *kept;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) to[i] = from[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) { counters[0] = cursor_at_start + n; counters[1] = hits_at_start + n; }
}
// the guides still active, packed: their longs and their numbers in the caller's guide array
__global__ void k_bound_compact(const uint64_t *__restrict__ guides, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, uint32_t n,
                                uint64_t *__restrict__ active, uint32_t *__restrict__ gmap) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n || !flag[g]) return;
    active[pos[g]] = guides[g];
    gmap[pos[g]] = g;
}

}  // namespace ffh
