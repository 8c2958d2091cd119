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
// Brute force over all (guide, target) pairs: every target is read once per 256 guides; a necessary condition on 13-14 bases
// of each pairing (see the kernel) rejects all but ~1e-3 of the pairs before the 36 bulge positions are tried.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ffh_kernels.hpp"

namespace ffh {

constexpr int kBulgeGuides = 256;  // guides staged in LDS per pass over the targets

// hit record: key = (guide << tbits) | database index (sorted on), payload = mismatches | type << 8 | position << 16
__global__ __launch_bounds__(256) void k_bulge_scan(const uint64_t *__restrict__ targets, uint64_t n_targets, const uint64_t *__restrict__ guides, uint32_t n_guides,
                                                    Geometry geo, int max_mm, int max_bulge, int tttv, int tbits, uint64_t *__restrict__ hit_key,
                                                    uint64_t *__restrict__ hit_val, unsigned long long *__restrict__ cursor, uint64_t cap) {
    __shared__ uint2 gk[kBulgeGuides];
    const uint64_t ti = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live_t = ti < n_targets;
    const uint64_t t = live_t ? targets[ti] : 0ull;
    const uint64_t pt = planar_key(t, geo.c0, geo.lc);
    const uint32_t Ht = (uint32_t)(pt >> 32), Lt = (uint32_t)pt, full = (1u << geo.lc) - 1u;
    // TTTV: the fourth PAM base (string index 3 of the 24-mer, bits 41:40) must not be T
    const bool pam_ok = !tttv || ((uint32_t)(t >> 40) & 3u) != 3u;
    // Necessary condition for a bulge at k to stay within max_mm, by zone of k (a = lc/3, b = 2 lc/3; head(k) = top k bits of the
    // unshifted pairing, tail(k) = bits 1 .. lc-1-k of a shifted one; both only grow towards their end of the protospacer):
    //   k <= a      : tail(k) contains tail(a)                                   -> popc(Dx & tail(a)) <= max_mm
    //   a < k <= b  : head(k) contains head(a+1), tail(k) contains tail(b)        -> popc(D0 & head(a+1)) + popc(Dx & tail(b)) <= max_mm
    //   k > b       : head(k) contains head(b+1)                                  -> popc(D0 & head(b+1)) <= max_mm
    // 13-14 bases each: ~1e-4 of random pairs pass one of them, so a 64-lane wave rarely enters the 36-position loop.
    const int lc = geo.lc, za = lc / 3, zb = (2 * lc) / 3;
    const uint32_t head_a = (((1u << (za + 1)) - 1u) << (lc - za - 1)) & full, head_b = (((1u << (zb + 1)) - 1u) << (lc - zb - 1)) & full;
    const uint32_t tail_a = ((1u << (lc - 1 - za)) - 1u) << 1, tail_b = ((1u << (lc - 1 - zb)) - 1u) << 1;
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
            const uint32_t D0 = (g.x ^ Ht) | (g.y ^ Lt);
            int best = __popc(D0), type = 0, pos = 0;
            if (max_bulge > 0) {
                const uint32_t D1 = ((((g.x << 1) ^ Ht) | ((g.y << 1) ^ Lt)) & full) & ~1u;  // bit 0 would pair g_20: not a base
                const uint32_t D2 = (((g.x ^ (Ht << 1)) | (g.y ^ (Lt << 1))) & full) & ~1u;  // bit 0 would pair t_20
                const int ha = __popc(D0 & head_a);
                const bool maybe = __popc(D0 & head_b) <= max_mm || __popc(D1 & tail_a) <= max_mm || __popc(D2 & tail_a) <= max_mm ||
                                   ha + __popc(D1 & tail_b) <= max_mm || ha + __popc(D2 & tail_b) <= max_mm;
                if (maybe) {
                    for (int k = 1; k <= lc - 2; ++k) {
                        const uint32_t head = D0 & ((((1u << k) - 1u) << (lc - k)) & full);
                        const uint32_t tailmask = ((1u << (lc - 1 - k)) - 1u) << 1;
                        const int rna = __popc(head) + __popc(D1 & tailmask);
                        const int dna = __popc(head) + __popc(D2 & tailmask);
                        if (rna < best) { best = rna; type = 1; pos = k; }
                        else if (rna == best && type == 2) { type = 1; pos = k; }  // RNA before DNA at equal cost (cannot lower k: k ascends)
                        if (dna < best) { best = dna; type = 2; pos = k; }
                    }
                }
            }
            const bool hit = live_t && pam_ok && best <= max_mm;
            const uint64_t m = __ballot(hit);
            if (m) {
                unsigned long long base = 0;
                if (lane_id() == 0) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
                base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
                if (hit) {
                    const unsigned long long slot = base + mbcnt(m);
                    if (slot < cap) {
                        hit_key[slot] = ((uint64_t)(g0 + j) << tbits) | ti;
                        hit_val[slot] = (uint64_t)best | ((uint64_t)type << 8) | ((uint64_t)pos << 16);
                    }
                }
            }
        }
    }
}

// sorted hits -> the arrays handed to the caller
__global__ void k_bulge_unpack(const uint64_t *__restrict__ key, const uint64_t *__restrict__ val, uint64_t n, int tbits, const uint64_t *__restrict__ targets,
                               uint64_t *__restrict__ out_target, uint8_t *__restrict__ out_mm, uint8_t *__restrict__ out_type, uint8_t *__restrict__ out_pos) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_target[i] = targets[key[i] & ((1ull << tbits) - 1ull)];
    const uint64_t v = val[i];
    out_mm[i] = (uint8_t)v; out_type[i] = (uint8_t)(v >> 8); out_pos[i] = (uint8_t)(v >> 16);
}

}  // namespace ffh
