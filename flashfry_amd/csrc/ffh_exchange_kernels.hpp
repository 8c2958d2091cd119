// ffh_exchange_kernels.hpp -- the device side of the shards' exchange (csrc/ffh_comm.hpp): the fold of the per-guide records in shard order
// and the kernels of the exchange by guide slices.  Kept free of everything but GuideSummary (ffh_kernels.hpp) and the HIP qualifiers, so
// that tests/exchange_emul_main.cpp can run the SAME source on the CPU (one thread after the other) and check the slice form's index
// arithmetic against the all-gather form on any box -- the transports above these kernels still need a GPU.
#pragma once
#include <stdint.h>

namespace ffh {

// The local half of the exchange: every rank holds every shard's per-guide aggregates (ONE all-gather of the 88-byte summaries, plus a
// status record per shard) and folds them itself, in shard order = database order:
//   * prior of shard r = the saturated position totals of the shards before it (CRISPRSiteOT.full, crispr/CRISPRSiteOT.scala:39-46,
//     continued across shards);
//   * a shard whose prior is 0, or whose prior + own total stays below maximumOffTargets, aggregated exactly what the unsharded run
//     keeps of its hits: its record is used as it is;
//   * a shard whose prior already reached the limit contributes nothing;
//   * the ONE shard per guide in which a non-zero prior and its own hits cross the limit kept too much: its record is left out and the
//     guide is counted in flag[0] -- the owner aggregates it again with the prior (k_guide_epilogue's fix-up form) and a second
//     round, adjusted = 1, folds the corrected records as they are.  A guide set without OVERFLOW guides (the benchmark's) never
//     needs the second round: one collective per step.
// Integer lanes add, overflow / cfd_max / jost_max take the maximum, the closest hit the minimum with its count summed over the
// shards at that level, the f64 sums are added in shard order (deterministic).  flag[0] bit 31: a shard reported a failure; bit 30: a
// failure OTHER than "more raw hits than one scan holds" (status word kStatusTooManyHits), after which every rank halves the guide set.
constexpr uint32_t kStatusTooManyHits = 0xFEFEFEFEu;   // (a status record is a memset: 0x00 fine, 0xFE this, 0xFF any other failure)
__global__ void k_exchange_reduce(const GuideSummary *__restrict__ all /* [world][n + 1] */, uint32_t n, uint32_t world, uint32_t clamp, int adjusted, uint32_t me,
                                  uint32_t *__restrict__ prior_out /* [n]: prior of shard `me` (first round only) */, GuideSummary *__restrict__ red, uint32_t *__restrict__ flag) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0)
        for (uint32_t r = 0; r < world; ++r) {
            const uint32_t status = all[(size_t)r * (n + 1) + n].n_hits;
            if (status != 0u) atomicOr(flag, status == kStatusTooManyHits ? 0x80000000u : 0xC0000000u);
        }
    if (g >= n) return;
    GuideSummary acc{};
    acc.closest = 0xFFFFFFFFu;
    uint64_t run = 0;
    bool crossing = false, any = false;
    for (uint32_t r = 0; r < world; ++r) {
        const GuideSummary v = all[(size_t)r * (n + 1) + g];
        bool use = true;
        if (!adjusted) {
            const uint32_t t = min(v.ot_count, clamp), p = (uint32_t)(run < clamp ? run : clamp);
            if (r == me) prior_out[g] = p;
            run += t;
            if (p > 0u && (uint64_t)p + t >= clamp) {
                use = false;
                if (p >= clamp) acc.overflow = 1u;   // (what the shard's own pass with this prior reports: full before its first hit)
                else crossing = true;
            }
        }
        if (!use) continue;
        acc.n_hits += v.n_hits; acc.ot_count += v.ot_count; acc.in_genome += v.in_genome; acc.n_scored += v.n_scored;
        for (int k = 0; k < 5; ++k) acc.hist[k] += v.hist[k];
        acc.overflow = max(acc.overflow, v.overflow);
        acc.cfd_max = fmax(acc.cfd_max, v.cfd_max); acc.jost_max = fmax(acc.jost_max, v.jost_max);
        if (v.closest < acc.closest) { acc.closest = v.closest; acc.closest_count = v.closest_count; }
        else if (v.closest == acc.closest && v.closest != 0xFFFFFFFFu) acc.closest_count += v.closest_count;
        if (!any) { acc.cfd_sum = v.cfd_sum; acc.hsu_sum = v.hsu_sum; acc.jost_sum = v.jost_sum; any = true; }
        else { acc.cfd_sum += v.cfd_sum; acc.hsu_sum += v.hsu_sum; acc.jost_sum += v.jost_sum; }
    }
    if (crossing) atomicAdd(flag, 1u);
    red[g] = acc;
}


// ---- the exchange by GUIDE SLICES (round 6; selectable: ffh_comm_set_exchange / FFH_EXCHANGE=slice) ----------------------------------
// The all-gather form hands every rank every shard's record of every guide: world x G x 88 bytes received per rank (70 MB at world 8),
// of which a rank needs, to fold, nothing but ... everything, because every rank folds every guide.  The slice form lets rank j fold only
// the guides of slice j = [j * sl, (j + 1) * sl), sl = ceil(G / world):
//   (1) all-to-all: shard i sends shard j its records of slice j + its status record           G x 88 x (world - 1) / world bytes per rank
//   (2) rank j folds its slice in shard order (the arithmetic of k_exchange_reduce) and works out, for EVERY shard, the prior of the
//       slice's guides
//   (3) all-to-all back: the priors (4 bytes per guide and shard) + the slice's flag word; every shard assembles its own prior[G] -- what
//       the second round and ffh_comm_shard_lists need -- and the sum of the flag words
//   (4) all-gather of the folded slices: every rank holds the reduced aggregates, as in the all-gather form          G x 88 bytes per rank
// Three collectives instead of one, an eighth of the payload at world 8.  Results are bit-identical to the all-gather form: the same
// records folded by the same code in the same order.
__global__ void k_slice_pack(const GuideSummary *__restrict__ summ /* [G + 1], record G = status */, uint32_t G, uint32_t sl, uint32_t world,
                             GuideSummary *__restrict__ send /* [world][sl + 1] */) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= world * (sl + 1u)) return;
    const uint32_t j = t / (sl + 1u), k = t % (sl + 1u), g = j * sl + k;
    // (moved as eleven 64-bit words: a GuideSummary held in a local was 96 bytes of scratch per lane)
    static_assert(sizeof(GuideSummary) == 11 * sizeof(uint64_t), "the record is moved as eleven 64-bit words");
    const uint64_t *src = k == sl ? (const uint64_t *)(summ + G) : g < G ? (const uint64_t *)(summ + g) : nullptr;   // (past G: the padding of the last slices)
    uint64_t *dst = (uint64_t *)(send + t);
#pragma unroll
    for (int w = 0; w < 11; ++w) dst[w] = src ? src[w] : 0ull;
}
// fold of one slice: `all` = [world][sl + 1] (per source shard: the slice's records, then the shard's status record); n_slice guides of it
// exist.  prior_all [world][sl + 1]: row r = the prior of shard r for the slice's guides (first round only; word sl of every row is filled
// with the slice's flag word by k_slice_flag).  flag[0] as in k_exchange_reduce.
__global__ void k_exchange_reduce_slice(const GuideSummary *__restrict__ all, uint32_t n_slice, uint32_t sl, uint32_t world, uint32_t clamp, int adjusted,
                                        uint32_t *__restrict__ prior_all, GuideSummary *__restrict__ red /* [sl] */, uint32_t *__restrict__ flag) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0)
        for (uint32_t r = 0; r < world; ++r) {
            const uint32_t status = all[(size_t)r * (sl + 1) + sl].n_hits;
            if (status != 0u) atomicOr(flag, status == kStatusTooManyHits ? 0x80000000u : 0xC0000000u);
        }
    if (k >= sl) return;
    GuideSummary acc{};
    acc.closest = 0xFFFFFFFFu;
    if (k >= n_slice) {   // (padding of the last slices: a defined record, a zero prior)
        red[k] = acc;
        if (!adjusted) for (uint32_t r = 0; r < world; ++r) prior_all[(size_t)r * (sl + 1) + k] = 0u;
        return;
    }
    uint64_t run = 0;
    bool crossing = false, any = false;
    for (uint32_t r = 0; r < world; ++r) {
        const GuideSummary v = all[(size_t)r * (sl + 1) + k];
        bool use = true;
        if (!adjusted) {
            const uint32_t t = min(v.ot_count, clamp), p = (uint32_t)(run < clamp ? run : clamp);
            prior_all[(size_t)r * (sl + 1) + k] = p;
            run += t;
            if (p > 0u && (uint64_t)p + t >= clamp) {
                use = false;
                if (p >= clamp) acc.overflow = 1u;
                else crossing = true;
            }
        }
        if (!use) continue;
        acc.n_hits += v.n_hits; acc.ot_count += v.ot_count; acc.in_genome += v.in_genome; acc.n_scored += v.n_scored;
        for (int q = 0; q < 5; ++q) acc.hist[q] += v.hist[q];
        acc.overflow = max(acc.overflow, v.overflow);
        acc.cfd_max = fmax(acc.cfd_max, v.cfd_max); acc.jost_max = fmax(acc.jost_max, v.jost_max);
        if (v.closest < acc.closest) { acc.closest = v.closest; acc.closest_count = v.closest_count; }
        else if (v.closest == acc.closest && v.closest != 0xFFFFFFFFu) acc.closest_count += v.closest_count;
        if (!any) { acc.cfd_sum = v.cfd_sum; acc.hsu_sum = v.hsu_sum; acc.jost_sum = v.jost_sum; any = true; }
        else { acc.cfd_sum += v.cfd_sum; acc.hsu_sum += v.hsu_sum; acc.jost_sum += v.jost_sum; }
    }
    if (crossing) atomicAdd(flag, 1u);
    red[k] = acc;
}
// the slice's flag word into word sl of every row of prior_all (it travels back with the priors)
__global__ void k_slice_flag(const uint32_t *__restrict__ flag, uint32_t sl, uint32_t world, uint32_t *__restrict__ prior_all) {
    if (threadIdx.x < world) prior_all[(size_t)threadIdx.x * (sl + 1) + sl] = flag[0];
}
// what came back: prior_in [world][sl + 1], row j = my priors of slice j + slice j's flag word -> prior[G], flag[0] = failure bits OR-ed,
// crossing counts added
__global__ void k_slice_assemble(const uint32_t *__restrict__ prior_in, uint32_t G, uint32_t sl, uint32_t world, int with_prior, uint32_t *__restrict__ prior, uint32_t *__restrict__ flag) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) {
        uint32_t bits = 0u, count = 0u;
        for (uint32_t j = 0; j < world; ++j) { const uint32_t w = prior_in[(size_t)j * (sl + 1) + sl]; bits |= w & 0xC0000000u; count += w & 0x3FFFFFFFu; }
        flag[0] = bits | (count > 0x3FFFFFFFu ? 0x3FFFFFFFu : count);
    }
    if (with_prior && g < G) prior[g] = prior_in[(size_t)(g / sl) * (sl + 1) + g % sl];
}

}  // namespace ffh
