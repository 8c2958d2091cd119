// ffh_index.hpp -- `index` on the device: target-site discovery, sort and de-duplication.
//
// Replaces, for a whole genome at once,
//   SimpleSiteFinder            reference/ReferenceEncoder.scala:104-175   (PAM regex on both strands, every position)
//   BinWriter.addHit + the per-bin sort/merge of BlockReader.loadBlock    crispr/BinWriter.scala:58-100,
//                                                                          reference/binary/BlockReader.scala:54-159
// The regexes are one consumed character plus a look-ahead, so every position is tested independently: a predicate over
// `scan length` bases with a set of allowed bases per offset.  Discovery order (contig by contig, all forward sites by
// start, then all reverse sites by start) is kept by counting first and writing at scanned offsets; the stable LSD sort
// on the sequence then leaves the positions of one target in discovery order.  That is a deliberate deterministic choice, not
// reference parity: the reference orders a bin with scala.util.Sorting.quickSort on the bases alone (CRISPRSite.compare,
// BlockReader.loadBlock), which is unstable above 16 elements, so for a multi-copy target neither its position order nor WHICH
// 32767 positions survive the Short.MaxValue cut is defined there.  Comparisons with real reference output must treat the
// position list of a target as a multiset (the oracle makes the same stable choice, so oracle parity is exact).
// Included by ffh_api.hip (single translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ffh_prims.hpp"

namespace ffh {

struct SitePattern {
    uint8_t fwd[24];  // allowed bases at offset k of a forward site: bit 0 = A, 1 = C, 2 = G, 3 = T
    uint8_t rev[24];  // the same for the pattern that marks a reverse-strand site on the forward strand
    int len;          // ParameterPack.totalScanLength
};

constexpr int kSiteThreads = 256;
constexpr int kSitePer = 4;
constexpr int kSiteTile = kSiteThreads * kSitePer;

__device__ __forceinline__ uint32_t base_code(uint8_t c) {  // the reference upper-cases the contig first (ReferenceEncoder.scala:63)
    c &= 0xDF;
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

// EMIT = false: blk_cnt[b] = forward sites of tile b, blk_cnt[nblocks + b] = reverse sites.
// EMIT = true : blk_off = exclusive scan of blk_cnt; sites go to keys/pos[site_base + ...] in discovery order.
template <bool EMIT>
__global__ __launch_bounds__(kSiteThreads) void k_site_scan(const uint8_t *__restrict__ seq, uint64_t n, SitePattern pat, uint32_t contig_id, uint32_t nblocks,
                                                            uint32_t *__restrict__ blk_cnt, const uint64_t *__restrict__ blk_off, uint64_t site_base,
                                                            uint64_t *__restrict__ keys, uint64_t *__restrict__ pos) {
    __shared__ uint8_t code[kSiteTile + 32];
    __shared__ uint32_t lds[8];
    const uint64_t base = (uint64_t)blockIdx.x * kSiteTile;
    for (uint32_t i = threadIdx.x; i < kSiteTile + 24; i += kSiteThreads) code[i] = base + i < n ? (uint8_t)base_code(seq[base + i]) : (uint8_t)4;
    __syncthreads();
    const int L = pat.len;
    const uint32_t t0 = threadIdx.x * kSitePer;
    uint32_t okf = (1u << kSitePer) - 1, okr = okf;  // bit j: start t0 + j still matches
    for (int k = 0; k < L; ++k) {
        const uint32_t mf = pat.fwd[k], mr = pat.rev[k];
#pragma unroll
        for (int j = 0; j < kSitePer; ++j) {
            const uint32_t c = code[t0 + j + k];
            okf &= ~((((mf >> c) & 1u) ^ 1u) << j);  // code 4 (not ACGT) is in no mask
            okr &= ~((((mr >> c) & 1u) ^ 1u) << j);
        }
    }
    const uint32_t cf = __popc(okf), cr = __popc(okr);
    uint32_t tot_f, tot_r;
    const uint32_t off_f = block_exclusive_scan<uint32_t>(cf, lds, tot_f);
    const uint32_t off_r = block_exclusive_scan<uint32_t>(cr, lds, tot_r);
    if (!EMIT) {
        if (threadIdx.x == 0) { blk_cnt[blockIdx.x] = tot_f; blk_cnt[nblocks + blockIdx.x] = tot_r; }
        return;
    }
    const uint64_t meta = ((uint64_t)L << 52) | ((uint64_t)contig_id << 32);  // BitPosition.encode, bitcoding/BitPosition.scala:51-63
    uint64_t wf = site_base + blk_off[blockIdx.x] + off_f, wr = site_base + blk_off[nblocks + blockIdx.x] + off_r;
#pragma unroll
    for (int j = 0; j < kSitePer; ++j) {
        const uint64_t start = base + t0 + j;
        if ((okf >> j) & 1u) {
            uint64_t e = 0;
            for (int k = 0; k < L; ++k) e = (e << 2) | code[t0 + j + k];  // BitEncoding.bitEncodeString :46-67, first base most significant
            keys[wf] = e;
            pos[wf] = meta | start;
            ++wf;
        }
        if ((okr >> j) & 1u) {
            uint64_t e = 0;
            for (int k = L - 1; k >= 0; --k) e = (e << 2) | (3u - code[t0 + j + k]);  // Utils.reverseCompString of the matched window
            keys[wr] = e;
            pos[wr] = meta | (1ull << 60) | start;
            ++wr;
        }
    }
}

// ---- de-duplication of the sorted sites: BlockReader.scala:138-159 ------------------------------------------------------
__global__ __launch_bounds__(256) void k_run_heads(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// rank = exclusive scan of head: the run that site i belongs to is rank[i] + head[i] - 1
__global__ __launch_bounds__(256) void k_run_starts(const uint32_t *__restrict__ head, const uint32_t *__restrict__ rank, uint64_t n, uint32_t n_runs,
                                                    uint32_t *__restrict__ start) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && head[i]) start[rank[i]] = (uint32_t)i;
    if (i == 0) start[n_runs] = (uint32_t)n;
}

constexpr uint32_t kMaxCount = 32767;  // Short.MaxValue: cap of the count AND of the position list (:147-153)

__global__ __launch_bounds__(256) void k_run_targets(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ start, uint32_t n_runs,
                                                     uint64_t *__restrict__ targets, uint32_t *__restrict__ count) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_runs) return;
    const uint32_t len = start[t + 1] - start[t];
    const uint32_t c = len < kMaxCount ? len : kMaxCount;
    count[t] = c;
    targets[t] = keys[start[t]] | ((uint64_t)c << 48);
}

__global__ __launch_bounds__(256) void k_run_positions(const uint64_t *__restrict__ pos, const uint32_t *__restrict__ head, const uint32_t *__restrict__ rank,
                                                       const uint32_t *__restrict__ start, const uint64_t *__restrict__ pos_off, uint64_t n,
                                                       uint64_t *__restrict__ positions) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = rank[i] + head[i] - 1u;
    const uint64_t r = i - start[t];
    if (r < kMaxCount) positions[pos_off[t] + r] = pos[i];
}

}  // namespace ffh
