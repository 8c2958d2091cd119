// ffh_bulge.hpp -- config C5 of BASELINE.json: Cas12a (Cpf1, 5' PAM TTTN/TTTV, 24-mer) off-target search with up to
// max_mismatch mismatches AND one bulge of one base.  The reference has no counterpart (no bulge / gap code anywhere,
// SURVEY.md section 8f-4), so the specification is this repository's; the tests check the kernel against a
// brute-force string restatement of that specification.
//
// Alignments of a guide protospacer g[0..19] (position 0 = next to the PAM) with a database target's protospacer t[0..19]:
//   none        g_i ~ t_i, i = 0..19                                   (BitEncoding.mismatches, the existing scan)
//   RNA bulge k guide base k is unpaired, 1 <= k <= 18:  g_i ~ t_i for i < k,  g_i ~ t_{i-1} for i > k      (19 pairs)
//   DNA bulge k target base k is unpaired, 1 <= k <= 18: g_i ~ t_i for i < k,  g_i ~ t_{i+1} for k <= i <= 18 (19 pairs; the
//               database stores 24-mers, so the genomic base that would pair with g_19 is not available and g_19 is left out)
// The best alignment is the one with the fewest mismatches, ties broken none < RNA < DNA, then smallest k; the pair is a
// hit if that count is <= max_mismatch.  In the planar domain (bit 19-i of a plane = base i) the three pairings are
//   D0 = (Hg ^ Ht) | (Lg ^ Lt)             D1 = ((Hg << 1) ^ Ht) | ((Lg << 1) ^ Lt)            D2 = (Hg ^ (Ht << 1)) | (Lg ^ (Lt << 1))
// and bulge k costs popc(D0 & top k bits) + popc(D1 or D2 & bits 1 .. 19-k).
// Two searches: candidates seeded from the resident scan images (k_bulge_seed, the default) and brute force over all pairs
// (k_bulge_scan, FFH_BULGE_BRUTE_FORCE: the seeded search's checker); both evaluate a pair with bulge_best.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ffh_kernels.hpp"

namespace ffh {

constexpr int kBulgeGuides = 256;  // guides staged in LDS per pass over the targets (brute-force kernel)

// masks of the necessary condition (see k_bulge_scan) for one protospacer length
struct BulgeMasks {
    uint32_t full, head_a, head_b, tail_a, tail_b;
    int lc;
};
__device__ __forceinline__ BulgeMasks bulge_masks(int lc) {
    BulgeMasks m;
    const int za = lc / 3, zb = (2 * lc) / 3;
    m.lc = lc;
    m.full = (1u << lc) - 1u;
    m.head_a = (((1u << (za + 1)) - 1u) << (lc - za - 1)) & m.full;
    m.head_b = (((1u << (zb + 1)) - 1u) << (lc - zb - 1)) & m.full;
    m.tail_a = ((1u << (lc - 1 - za)) - 1u) << 1;
    m.tail_b = ((1u << (lc - 1 - zb)) - 1u) << 1;
    return m;
}

// best alignment of one (guide, target) pair of planar keys: fewest mismatches, ties none < RNA < DNA, then smallest k.
// Necessary condition for a bulge at k to stay within max_mm, by zone of k (a = lc/3, b = 2 lc/3; head(k) = top k bits of the
// unshifted pairing, tail(k) = bits 1 .. lc-1-k of a shifted one; both only grow towards their end of the protospacer):
//   k <= a      : tail(k) contains tail(a)                                   -> popc(Dx & tail(a)) <= max_mm
//   a < k <= b  : head(k) contains head(a+1), tail(k) contains tail(b)        -> popc(D0 & head(a+1)) + popc(Dx & tail(b)) <= max_mm
//   k > b       : head(k) contains head(b+1)                                  -> popc(D0 & head(b+1)) <= max_mm
// 13-14 bases each: ~1e-4 of random pairs pass one of them, so a 64-lane wave rarely enters the 36-position loop.
__device__ __forceinline__ void bulge_best(uint32_t Hg, uint32_t Lg, uint32_t Ht, uint32_t Lt, const BulgeMasks &M, int max_mm, int max_bulge, int &best, int &type,
                                           int &pos) {
    const uint32_t D0 = (Hg ^ Ht) | (Lg ^ Lt);
    best = __popc(D0); type = 0; pos = 0;
    if (max_bulge <= 0) return;
    const uint32_t D1 = ((((Hg << 1) ^ Ht) | ((Lg << 1) ^ Lt)) & M.full) & ~1u;  // bit 0 would pair g_20: not a base
    const uint32_t D2 = (((Hg ^ (Ht << 1)) | (Lg ^ (Lt << 1))) & M.full) & ~1u;  // bit 0 would pair t_20
    const int ha = __popc(D0 & M.head_a);
    const bool maybe = __popc(D0 & M.head_b) <= max_mm || __popc(D1 & M.tail_a) <= max_mm || __popc(D2 & M.tail_a) <= max_mm ||
                       ha + __popc(D1 & M.tail_b) <= max_mm || ha + __popc(D2 & M.tail_b) <= max_mm;
    if (!maybe) return;
    const int lc = M.lc;
    for (int k = 1; k <= lc - 2; ++k) {
        const uint32_t head = D0 & ((((1u << k) - 1u) << (lc - k)) & M.full);
        const uint32_t tailmask = ((1u << (lc - 1 - k)) - 1u) << 1;
        const int rna = __popc(head) + __popc(D1 & tailmask);
        const int dna = __popc(head) + __popc(D2 & tailmask);
        if (rna < best) { best = rna; type = 1; pos = k; }
        else if (rna == best && type == 2) { type = 1; pos = k; }  // RNA before DNA at equal cost (cannot lower k: k ascends)
        if (dna < best) { best = dna; type = 2; pos = k; }
    }
}

// one wave appends its lanes' hits: one atomic per wave and call
__device__ __forceinline__ void bulge_emit(bool hit, uint64_t key, uint64_t val, uint64_t *__restrict__ hit_key, uint64_t *__restrict__ hit_val,
                                           unsigned long long *__restrict__ cursor, uint64_t cap) {
    const uint64_t m = __ballot(hit);
    if (!m) return;
    unsigned long long base = 0;
    if (lane_id() == 0) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
    base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
    if (hit) {
        const unsigned long long slot = base + mbcnt(m);
        if (slot < cap) { hit_key[slot] = key; hit_val[slot] = val; }
    }
}

// hit record: key = (guide << tbits) | database index (sorted on), payload = mismatches | type << 8 | position << 16
// Brute force over all (guide, target) pairs (FFH_BULGE_BRUTE_FORCE: the checker of the seeded search below): every target is
// read once per 256 guides.
__global__ __launch_bounds__(256) void k_bulge_scan(const uint64_t *__restrict__ targets, uint64_t n_targets, const uint64_t *__restrict__ guides, uint32_t n_guides,
                                                    Geometry geo, int max_mm, int max_bulge, int tttv, int tbits, uint64_t *__restrict__ hit_key,
                                                    uint64_t *__restrict__ hit_val, unsigned long long *__restrict__ cursor, uint64_t cap) {
    __shared__ uint2 gk[kBulgeGuides];
    const uint64_t ti = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live_t = ti < n_targets;
    const uint64_t t = live_t ? targets[ti] : 0ull;
    const uint64_t pt = planar_key(t, geo.c0, geo.lc);
    const uint32_t Ht = (uint32_t)(pt >> 32), Lt = (uint32_t)pt;
    // TTTV: the fourth PAM base (string index 3 of the 24-mer, bits 41:40) must not be T
    const bool pam_ok = !tttv || ((uint32_t)(t >> 40) & 3u) != 3u;
    const BulgeMasks M = bulge_masks(geo.lc);
    for (uint32_t g0 = 0; g0 < n_guides; g0 += kBulgeGuides) {
        __syncthreads();
        if (g0 + threadIdx.x < n_guides) {
            const uint64_t pg = planar_key(guides[g0 + threadIdx.x], geo.c0, geo.lc);
            gk[threadIdx.x] = make_uint2((uint32_t)(pg >> 32), (uint32_t)pg);
        }
        __syncthreads();
        const uint32_t ng = min((uint32_t)kBulgeGuides, n_guides - g0);
        for (uint32_t j = 0; j < ng; ++j) {
            const uint2 g = gk[j];
            int best, type, pos;
            bulge_best(g.x, g.y, Ht, Lt, M, max_mm, max_bulge, best, type, pos);
            const bool hit = live_t && pam_ok && best <= max_mm;
            bulge_emit(hit, ((uint64_t)(g0 + j) << tbits) | ti, (uint64_t)best | ((uint64_t)type << 8) | ((uint64_t)pos << 16), hit_key, hit_val, cursor, cap);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The seeded search: candidates from the two resident scan images (prefix buckets of width a, suffix buckets of width
// s = lc - a) instead of all targets.  An alignment within max_mm mismatches keeps <= max_mm of them inside any window of it, and
// on one side of the bulge the pairing is a plain (shifted or unshifted) comparison of consecutive bases, i.e. a bucket key:
//   P  no bulge, or a bulge at k >= a : target bases 0..a-1 pair with guide bases 0..a-1        -> prefix bucket within max_mm of the
//                                       guide's own prefix key
//   D  DNA bulge at k <= a-1          : target bases a..19 pair with guide bases a-1..18        -> suffix bucket within max_mm of the
//                                       guide's planes shifted right by one base
//   R  RNA bulge at k <= a            : target bases a..18 pair with guide bases a+1..19, target base 19 is unpaired
//                                       -> suffix bucket within max_mm of the guide's planes shifted left by one base, over the s - 1
//                                       paired bases, times the four values of the unpaired one
// Every k in 1..18 falls under P or under D / R, so the union of the three candidate sets holds every hit; each candidate is then
// evaluated in full (bulge_best), which makes the record independent of the seed that found it -- a pair found by two seeds
// appears twice with the same record and is merged after the sort (k_bulge_flag_unique).  The candidates' planes are rebuilt
// from the image itself (bucket id + the bit-sliced rest key of the slot), so the targets array is only touched for the TTTV check
// of actual hits.  One wave per (guide, 64 patterns of one seed).
// ---------------------------------------------------------------------------------------------------------------------
struct BulgeSeed {
    const uint32_t *bstart, *gstart, *gwords, *tidx, *patterns;
    uint32_t n_pat;
    int width, rest, gw;  // bucket width in bases, bases in the rest key, words per group
    int kind;             // 0 P, 1 D, 2 R
};

__global__ __launch_bounds__(256) void k_bulge_seed(BulgeSeed S, const uint64_t *__restrict__ targets, const uint64_t *__restrict__ guides /* of this launch */,
                                                    uint32_t n_guides, uint32_t guide_base /* number of guides[0] in the caller's array */, Geometry geo,
                                                    int max_mm, int max_bulge, int tttv, int tbits, uint64_t *__restrict__ hit_key, uint64_t *__restrict__ hit_val,
                                                    unsigned long long *__restrict__ cursor, uint64_t cap) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t slices = (S.n_pat + 63u) >> 6;
    const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t g = (uint32_t)(wid / slices), sl = (uint32_t)(wid % slices);
    if (g >= n_guides) return;
    const int lc = geo.lc, w = S.width;
    const uint64_t pg = planar_key(guides[g], geo.c0, lc);
    const uint32_t Hg = (uint32_t)(pg >> 32), Lg = (uint32_t)pg, wm = (1u << w) - 1u;
    uint32_t qh, ql;
    if (S.kind == 0) { qh = Hg >> (lc - w); ql = Lg >> (lc - w); }
    else if (S.kind == 1) { qh = (Hg >> 1) & wm; ql = (Lg >> 1) & wm; }
    else { qh = (Hg << 1) & wm; ql = (Lg << 1) & wm; }
    const uint32_t q = (qh << w) | ql;
    const uint32_t p = sl * 64 + lane;
    uint32_t b = 0, k0 = 0, nt = 0, g0 = 0;
    if (p < S.n_pat) {
        b = q ^ S.patterns[p];
        k0 = S.bstart[b]; nt = S.bstart[b + 1] - k0; g0 = S.gstart[b];
    }
    (void)k0;
    const BulgeMasks M = bulge_masks(lc);
    uint64_t live = __ballot(nt > 0);
    while (live) {
        const int src = __builtin_ctzll(live);
        live &= live - 1;
        const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)b, src), n = (uint32_t)__builtin_amdgcn_readlane((int)nt, src);
        const uint32_t gg0 = (uint32_t)__builtin_amdgcn_readlane((int)g0, src);
        const uint32_t bh = bb >> w, bl = bb & wm;
        for (uint32_t c = 0; c < n; c += 64) {
            const uint32_t k = c + lane;
            const bool ok = k < n;
            const uint32_t *gw = S.gwords + (size_t)(gg0 + (k >> 5)) * S.gw;
            const uint32_t slot = k & 31u;
            uint32_t rh = 0, rl = 0;
            if (ok)
                for (int i = 0; i < S.rest; ++i) {
                    rh |= ((gw[2 * i] >> slot) & 1u) << i;
                    rl |= ((gw[2 * i + 1] >> slot) & 1u) << i;
                }
            // prefix image: the bucket holds the first a bases (high bits), the rest key the others; suffix image the other way round
            const uint32_t Ht = S.kind == 0 ? ((bh << (lc - w)) | rh) : ((rh << w) | bh);
            const uint32_t Lt = S.kind == 0 ? ((bl << (lc - w)) | rl) : ((rl << w) | bl);
            int best, type, pos;
            bulge_best(Hg, Lg, Ht, Lt, M, max_mm, max_bulge, best, type, pos);
            bool hit = ok && best <= max_mm;
            uint32_t ti = 0;
            if (hit) {
                ti = S.tidx[(size_t)gg0 * 32 + k];
                if (tttv && ((uint32_t)(targets[ti] >> 40) & 3u) == 3u) hit = false;
            }
            bulge_emit(hit, ((uint64_t)(guide_base + g) << tbits) | ti, (uint64_t)best | ((uint64_t)type << 8) | ((uint64_t)pos << 16), hit_key, hit_val, cursor, cap);
        }
    }
}

// sorted hits: 1 where a (guide, target) key differs from the one before it
__global__ void k_bulge_flag_unique(const uint64_t *__restrict__ key, uint64_t n, uint32_t *__restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}

// sorted hits -> the arrays handed to the caller (first record of every key; dst = exclusive scan of the flags)
__global__ void k_bulge_unpack(const uint64_t *__restrict__ key, const uint64_t *__restrict__ val, uint64_t n, int tbits, const uint64_t *__restrict__ targets,
                               const uint32_t *__restrict__ flag, const uint64_t *__restrict__ dst, uint64_t *__restrict__ out_key, uint64_t *__restrict__ out_target,
                               uint8_t *__restrict__ out_mm, uint8_t *__restrict__ out_type, uint8_t *__restrict__ out_pos) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint64_t o = dst[i];
    out_key[o] = key[i];
    out_target[o] = targets[key[i] & ((1ull << tbits) - 1ull)];
    const uint64_t v = val[i];
    out_mm[o] = (uint8_t)v; out_type[o] = (uint8_t)(v >> 8); out_pos[o] = (uint8_t)(v >> 16);
}

}  // namespace ffh
