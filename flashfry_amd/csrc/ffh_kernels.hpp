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
                             uint32_t *__restrict__ gbucket) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint64_t pk = planar_key(guides[g], geo.c0, geo.lc);
    gkey[g] = pk;
    gbucket[g] = SUFFIX ? suffix_bucket(pk, width) : prefix_bucket(pk, geo.lc, width);
}

__global__ void k_item_count(const uint32_t *__restrict__ gbucket, uint32_t n_guides, const uint32_t *__restrict__ patterns, uint32_t n_pat,
                             const uint32_t *__restrict__ bstart, uint32_t *__restrict__ icount) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_guides * n_pat) return;
    const uint32_t g = (uint32_t)(i / n_pat), j = (uint32_t)(i % n_pat);
    const uint32_t b = gbucket[g] ^ patterns[j];
    if (bstart[b + 1] != bstart[b]) atomicAdd(&icount[b], 1u);
}

__global__ void k_item_fill(const uint32_t *__restrict__ gbucket, const uint64_t *__restrict__ gkey, uint32_t n_guides,
                            const uint32_t *__restrict__ patterns, uint32_t n_pat, const uint32_t *__restrict__ bstart,
                            const uint32_t *__restrict__ istart, uint32_t *__restrict__ ifill, const uint32_t *__restrict__ item_base,
                            uint64_t *__restrict__ item_key, uint32_t *__restrict__ item_gid) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_guides * n_pat) return;
    const uint32_t g = (uint32_t)(i / n_pat), j = (uint32_t)(i % n_pat);
    const uint32_t b = gbucket[g] ^ patterns[j];
    if (bstart[b + 1] == bstart[b]) return;
    const uint32_t pos = *item_base + istart[b] + atomicAdd(&ifill[b], 1u);
    item_key[pos] = gkey[g];
    item_gid[pos] = g;
}

constexpr int kTileTargets = 64;  // one target per lane

__global__ void k_tile_count(const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ istart, uint32_t n_buckets,
                             uint32_t *__restrict__ tcount, unsigned long long *__restrict__ pairs) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mine = 0;
    if (b < n_buckets) {
        const uint32_t nt = bstart[b + 1] - bstart[b], ng = istart[b + 1] - istart[b];
        tcount[b] = ng ? (nt + kTileTargets - 1) / kTileTargets : 0;
        mine = (unsigned long long)nt * ng;
    }
    // wave-reduce the pair count, one atomic per wave
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(pairs, mine);
}

// tile = {first key, #keys | side << 31, first item, #items}; tiles of both images share one list so that ONE
// compare launch covers the prefix and the suffix pass
__global__ void k_tile_fill(const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ istart, const uint32_t *__restrict__ tstart,
                            uint32_t n_buckets, const uint32_t *__restrict__ item_base, const uint32_t *__restrict__ tile_base, uint32_t side,
                            uint4 *__restrict__ tiles) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    const uint32_t t0 = tstart[b], nt = tstart[b + 1] - t0;
    if (!nt) return;
    const uint32_t k0 = bstart[b], kn = bstart[b + 1] - k0, g0 = *item_base + istart[b], gn = istart[b + 1] - istart[b];
    uint4 *out = tiles + *tile_base + t0;
    for (uint32_t c = 0; c < nt; ++c) {
        const uint32_t kb = c * kTileTargets;
        out[c] = make_uint4(k0 + kb, min(kn - kb, (uint32_t)kTileTargets) | (side << 31), g0, gn);
    }
}

// ---------------------------------------------------------------------------------------------------------
// THE HOT KERNEL.  One wave owns one tile at a time: <= 64 bucket-mates (one target per lane, streamed coalesced
// from the scan image) against that bucket's candidate guides (wave-uniform, fetched through the scalar cache).
// Per (guide, target): 2 x v_xor, v_or, v_bcnt, v_cmp.  Hits are compacted with ballot + mbcnt into a per-wave
// LDS staging buffer and flushed with ONE global atomic per ~200 hits (a single global cursor saturates at
// < 1e8 atomics/s, far below the hit rate).
//   Suffix-image tiles: the same pair can only also be found through the prefix image when its prefix part has
//   <= r1 mismatches, so it is emitted from a suffix tile only if the prefix part has MORE than r1.
// ---------------------------------------------------------------------------------------------------------
constexpr int kCmpThreads = 256;
constexpr int kStage = 256;  // staged hits per wave

__global__ __launch_bounds__(kCmpThreads) void k_compare(const uint4 *__restrict__ tiles, const uint32_t *__restrict__ n_tiles_a,
                                                         const uint32_t *__restrict__ n_tiles_b, const uint64_t *__restrict__ keys_p,
                                                         const uint32_t *__restrict__ tidx_p, const uint64_t *__restrict__ keys_s,
                                                         const uint32_t *__restrict__ tidx_s, const uint64_t *__restrict__ item_key,
                                                         const uint32_t *__restrict__ item_gid, int max_mm, uint32_t prefix_mask, int r1,
                                                         uint64_t *__restrict__ hits, unsigned long long *__restrict__ cursor, uint64_t cap) {
    __shared__ uint64_t stage[kCmpThreads / 64][kStage];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (kCmpThreads / 64);
    const uint32_t n_tiles = *n_tiles_a + *n_tiles_b;
    uint64_t *my = stage[wave];
    uint32_t fill = 0;

    auto flush = [&]() {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(cursor, (unsigned long long)fill);
        base = __shfl(base, 0, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane; i < fill; i += 64)
            if (base + i < cap) hits[base + i] = my[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        fill = 0;
    };
    auto emit = [&](bool hit, uint32_t gid_index, uint32_t ti, const uint32_t *__restrict__ ig) {
        const uint64_t mask = __ballot(hit);
        if (mask) {
            const uint32_t gid = ig[gid_index];
            if (hit) my[fill + mbcnt(mask)] = ((uint64_t)gid << 32) | ti;
            fill += (uint32_t)__popcll(mask);
            if (fill > kStage - 64) flush();
        }
    };

    for (uint32_t t = blockIdx.x * (kCmpThreads / 64) + wave; t < n_tiles; t += n_waves) {
        const uint4 tile = tiles[t];
        const bool suffix = (tile.y >> 31) != 0;  // wave-uniform
        const uint32_t nk = tile.y & 0x7FFFFFFFu;
        const bool valid = lane < nk;
        const uint64_t *__restrict__ keys = suffix ? keys_s : keys_p;
        const uint32_t *__restrict__ tidx = suffix ? tidx_s : tidx_p;
        const uint64_t k = valid ? keys[tile.x + lane] : 0;
        const uint32_t ti = valid ? tidx[tile.x + lane] : 0;
        const uint32_t kh = (uint32_t)(k >> 32), kl = (uint32_t)k;
        const uint64_t *__restrict__ ik = item_key + tile.z;
        const uint32_t *__restrict__ ig = item_gid + tile.z;
        const uint32_t ng = tile.w;
        if (!suffix) {
            for (uint32_t j = 0; j < ng; ++j) {
                const uint64_t g = ik[j];
                const uint32_t y = (kh ^ (uint32_t)(g >> 32)) | (kl ^ (uint32_t)g);
                emit(valid && (__popc(y) <= max_mm), j, ti, ig);
            }
        } else {
            for (uint32_t j = 0; j < ng; ++j) {
                const uint64_t g = ik[j];
                const uint32_t y = (kh ^ (uint32_t)(g >> 32)) | (kl ^ (uint32_t)g);
                emit(valid && (__popc(y) <= max_mm) && (__popc(y & prefix_mask) > r1), j, ti, ig);
            }
        }
    }
    if (fill) flush();
}

__global__ void k_add_u64(uint64_t *__restrict__ v, uint64_t n, uint64_t add) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] += add;
}

// ---------------------------------------------------------------------------------------------------------
// epilogue: ordered cut-off (crispr/CRISPRSiteOT.scala:39-46) + per-hit scores + per-guide aggregates
// ---------------------------------------------------------------------------------------------------------
__global__ void k_segments(const uint64_t *__restrict__ hits, uint64_t n, uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = (uint32_t)(hits[i] >> 32);
    if (i == 0 || (uint32_t)(hits[i - 1] >> 32) != g) seg_begin[g] = (uint32_t)i;
    if (i == n - 1 || (uint32_t)(hits[i + 1] >> 32) != g) seg_end[g] = (uint32_t)(i + 1);
}

// sum of positions over all hits of this shard, saturating (multi-GPU exchange)
__global__ void k_shard_totals(const uint64_t *__restrict__ hits, const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_end,
                               const uint64_t *__restrict__ targets, uint32_t n_guides, uint32_t clamp, uint32_t *__restrict__ totals) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_guides) return;
    uint32_t run = 0;
    for (uint32_t h = seg_begin[g]; h < seg_end[g] && run < clamp; ++h) run += (uint32_t)(targets[(uint32_t)hits[h]] >> 48);
    totals[g] = run < clamp ? run : clamp;
}

// a hit is kept iff the running total BEFORE it is < overflow; the total grows by the hit's position count
__global__ void k_cutoff(const uint64_t *__restrict__ hits, const uint32_t *__restrict__ seg_begin, const uint32_t *__restrict__ seg_end,
                         const uint64_t *__restrict__ targets, const uint32_t *__restrict__ prior, uint32_t n_guides, uint32_t overflow,
                         uint32_t *__restrict__ n_ret, uint32_t *__restrict__ ot_count, uint32_t *__restrict__ full) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_guides) return;
    const uint32_t p0 = prior ? prior[g] : 0;
    uint32_t run = p0, kept = 0;
    for (uint32_t h = seg_begin[g]; h < seg_end[g] && run < overflow; ++h) {
        run += (uint32_t)(targets[(uint32_t)hits[h]] >> 48);
        ++kept;
    }
    n_ret[g] = kept;
    ot_count[g] = run - p0;
    full[g] = run >= overflow;
}

struct ScoreTables {
    double cfd_mm[20 * 4 * 4];
    double cfd_pam[16];
    double hsu_coeff[20];
};

// per retained hit: target long, mismatches, position count, pam*CFD (Doench2016CFDScore.scala:67-73) and the
// Hsu2013 hit score (CrisprMitEduOffTarget.scala:107-148); both NaN for a 0-mismatch hit (the on-target itself).
__global__ void k_score_hits(const uint64_t *__restrict__ hits, uint64_t n_hits, const uint32_t *__restrict__ seg_begin,
                             const uint32_t *__restrict__ n_ret, const uint64_t *__restrict__ ret_off, const uint64_t *__restrict__ targets,
                             const uint64_t *__restrict__ guides, Geometry geo, const ScoreTables *__restrict__ tab, uint64_t *__restrict__ out_target,
                             uint8_t *__restrict__ out_mm, uint32_t *__restrict__ out_cnt, uint32_t *__restrict__ out_tidx,
                             double *__restrict__ out_cfd, double *__restrict__ out_hsu) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hits) return;
    const uint32_t g = (uint32_t)(hits[i] >> 32), ti = (uint32_t)hits[i];
    const uint32_t local = (uint32_t)i - seg_begin[g];
    if (local >= n_ret[g]) return;
    const uint64_t o = ret_off[g] + local;
    const uint64_t t = targets[ti], gd = guides[g];
    const uint64_t pg = planar_key(gd, geo.c0, geo.lc), pt = planar_key(t, geo.c0, geo.lc);
    const uint32_t y = ((uint32_t)(pg >> 32) ^ (uint32_t)(pt >> 32)) | ((uint32_t)pg ^ (uint32_t)pt);
    const int mm = __popc(y);
    out_target[o] = t;
    out_mm[o] = (uint8_t)mm;
    out_cnt[o] = (uint32_t)(t >> 48);
    out_tidx[o] = ti;
    double cfd = __builtin_nan(""), hsu = __builtin_nan("");
    if (geo.cas9_23 && mm != 0) {
        // base i (0 = 5' end) of a 23-mer sits at bits [2(22-i)+1 : 2(22-i)]
        double score = 1.0, part_one = 1.0;
        int first = -1, last = -1;
#pragma unroll
        for (int b = 0; b < 20; ++b) {
            const int sh = 2 * (22 - b);
            const uint32_t gb = (uint32_t)(gd >> sh) & 3u, ob = (uint32_t)(t >> sh) & 3u;
            score *= tab->cfd_mm[b * 16 + gb * 4 + ob];  // 1.0 where the bases agree
            if (gb != ob) {
                part_one = part_one * (1.0 - tab->hsu_coeff[b]);
                if (first < 0) first = b;
                last = b;
            }
        }
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
    out_cfd[o] = cfd;
    out_hsu[o] = hsu;
}

struct GuideSummary {  // mirrors ffh_guide_summary
    uint32_t n_hits, ot_count, overflow, hist[5], closest, closest_count, in_genome, n_scored;
    double cfd_max, cfd_sum, hsu_sum;
};

// one thread per guide walks its retained hits IN DATABASE ORDER so the f64 sums associate exactly like the
// reference's sequential folds (Doench2016CFDScore.scala:79, CrisprMitEduOffTarget.scala:104)
__global__ void k_guide_aggregate(const uint64_t *__restrict__ ret_off, const uint32_t *__restrict__ n_ret, const uint32_t *__restrict__ ot_count,
                                  const uint32_t *__restrict__ full, const uint8_t *__restrict__ mm, const uint32_t *__restrict__ cnt,
                                  const double *__restrict__ cfd, const double *__restrict__ hsu, uint32_t n_guides, GuideSummary *__restrict__ out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_guides) return;
    GuideSummary s;
    s.n_hits = n_ret[g]; s.ot_count = ot_count[g]; s.overflow = full[g];
    for (int k = 0; k < 5; ++k) s.hist[k] = 0;
    s.closest = 0xFFFFFFFFu; s.closest_count = 0; s.in_genome = 0; s.n_scored = 0;
    s.cfd_max = 0.0; s.cfd_sum = 0.0; s.hsu_sum = 0.0;
    const uint64_t b = ret_off[g];
    for (uint32_t k = 0; k < s.n_hits; ++k) {
        const uint32_t m = mm[b + k], c = cnt[b + k];
        if (m <= 4) s.hist[m] += c;                                        // ClosestHit.scala:57-59
        if (m < s.closest && m > 0) { s.closest = m; s.closest_count = c; } // :62-64
        else if (m == s.closest) s.closest_count += c;                      // :65-67
        if (m == 0) s.in_genome += c;                                       // DangerousSequences.scala:62
        const double f = cfd[b + k];
        if (f == f) {  // scored (not the on-target)
            s.cfd_sum += f * (double)c;
            if (s.n_scored == 0 || f > s.cfd_max) s.cfd_max = f;
            s.hsu_sum += hsu[b + k];
            s.n_scored++;
        }
    }
    out[g] = s;
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
