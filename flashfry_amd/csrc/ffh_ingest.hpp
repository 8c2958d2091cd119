// ffh_ingest.hpp -- decoding FlashFry bin payloads ON THE DEVICE.
//
// A bin payload is what DatabaseWriter.scala:58-111 puts into the BGZF body and what the traversers hand to
// BlockManager.compareBlock (blocks/BlockManager.scala:63-90):
//     [type]                                   1 = linear, 2 = indexed
//     [256 x (pos << 32 | size)]               indexed only: slice of every 4-base sub-bin, relative to the table end
//     { target long, count x position long }*  count = bits 63:48 of the target
// Whether a long is a target or a position is only known by walking the records, so the walk is the one sequential
// step -- but every sub-bin slice (indexed) / every bin (linear) starts at a record boundary, which gives millions of
// independent walks at genome scale.  The walks only MARK the targets; a scan of the marks turns the split into
// targets[] / positions[] into a fully coalesced copy.
//
//   k_block_heads   one thread per bin: block type, table validation (compareIndexedBlock :160-170), header length
//   k_block_walk    one thread per (bin, sub-bin): record walk of compareLinearBlock :225-252, marks[payload index] = 1
//   k_block_split   one thread per payload long: target -> targets[rank], position -> positions[index - rank]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ffh {

constexpr int kSubBins = 256;  // 4^4 sub-bins, BlockManager.scala:40-49

enum BlockError : uint32_t {
    kBlkEmpty = 1,          // zero-length payload (BlockManager.scala:72 would throw)
    kBlkShortTable = 2,     // indexed block shorter than its lookup table
    kBlkNotContiguous = 3,  // :167-168
    kBlkSliceRange = 4,
    kBlkCover = 5,
    kBlkType = 6,           // :85-87
    kBlkCount = 7,          // :232-233
    kBlkOverrun = 8,        // :235-236
};

// the first error in database order wins: key = (bin << 36) | (offset << 4) | code, smallest key kept
__device__ __forceinline__ void block_error(unsigned long long *err, uint32_t bin, uint64_t off, uint32_t code) {
    atomicMin(err, ((unsigned long long)bin << 36) | ((unsigned long long)(off & 0xFFFFFFFFull) << 4) | code);
}

__global__ __launch_bounds__(256) void k_block_heads(const int64_t *__restrict__ raw, const uint64_t *__restrict__ boff, uint32_t n_bins,
                                                     uint32_t *__restrict__ hdr, uint32_t *__restrict__ plen, unsigned long long *__restrict__ err) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins) return;
    const int64_t *blk = raw + boff[b];
    const uint64_t n = boff[b + 1] - boff[b];
    hdr[b] = 0; plen[b] = 0;
    if (n == 0) { block_error(err, b, 0, kBlkEmpty); return; }
    const int64_t type = blk[0];
    if (type == 1) { hdr[b] = 1; plen[b] = (uint32_t)(n - 1); return; }
    if (type != 2) { block_error(err, b, 0, kBlkType); return; }
    if (n - 1 < (uint64_t)kSubBins) { block_error(err, b, 0, kBlkShortTable); return; }
    const uint64_t payload = n - 1 - kSubBins;
    long long last_pos = 0, last_size = 0;
    uint64_t covered = 0;
    for (int i = 0; i < kSubBins; ++i) {
        const int64_t e = blk[1 + i];
        const int pos = (int)(e >> 32);
        const int size = (int)(uint32_t)e;
        if (last_pos != 0 && pos >= 0 && pos != last_pos + last_size) { block_error(err, b, (uint64_t)i, kBlkNotContiguous); return; }
        last_pos = pos > 0 ? pos : 0;
        last_size = size;
        if (pos >= 0 && size > 0) {
            if ((uint64_t)pos + (uint64_t)size > payload) { block_error(err, b, (uint64_t)i, kBlkSliceRange); return; }
            covered += (uint64_t)size;
        }
    }
    if (covered != payload) { block_error(err, b, (uint64_t)kSubBins, kBlkCover); return; }
    hdr[b] = 1 + kSubBins;
    plen[b] = (uint32_t)payload;
}

// pbase[b] = payload longs of the bins before b (exclusive scan of plen)
__global__ __launch_bounds__(256) void k_block_walk(const int64_t *__restrict__ raw, const uint64_t *__restrict__ boff, uint32_t n_bins,
                                                    const uint32_t *__restrict__ hdr, const uint32_t *__restrict__ plen, const uint64_t *__restrict__ pbase,
                                                    uint32_t *__restrict__ marks, unsigned long long *__restrict__ err) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = (uint32_t)(u >> 8), sub = (uint32_t)(u & 255u);
    if (b >= n_bins) return;
    const uint32_t h = hdr[b];
    uint64_t lo, hi;
    if (h == 0) return;  // refused by k_block_heads
    if (h == 1) {
        if (sub) return;
        lo = 0; hi = plen[b];
    } else {
        const int64_t e = raw[boff[b] + 1 + sub];
        const int pos = (int)(e >> 32), size = (int)(uint32_t)e;
        if (pos < 0 || size <= 0) return;
        lo = (uint64_t)pos; hi = lo + (uint64_t)size;
    }
    const int64_t *blk = raw + boff[b] + h;
    uint32_t *m = marks + pbase[b];
    for (uint64_t off = lo; off < hi;) {
        const int count = (int)(int16_t)((uint64_t)blk[off] >> 48);
        if (count <= 0) { block_error(err, b, off, kBlkCount); return; }
        if (hi < off + (uint64_t)count + 1) { block_error(err, b, off, kBlkOverrun); return; }
        m[off] = 1u;
        off += (uint64_t)count + 1;
    }
}

// rank[i] = marked longs before payload long i (exclusive scan of marks; rank[n] = number of targets)
__global__ __launch_bounds__(256) void k_block_split(const int64_t *__restrict__ raw, const uint64_t *__restrict__ boff, uint32_t n_bins,
                                                     const uint32_t *__restrict__ hdr, const uint64_t *__restrict__ pbase, const uint32_t *__restrict__ rank,
                                                     uint64_t n_payload, uint64_t *__restrict__ targets, uint64_t *__restrict__ positions) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_payload) return;
    uint32_t lo = 0, hi = n_bins;  // last bin whose payload starts at or before i (empty bins share a start: take the last)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pbase[mid] <= i) lo = mid; else hi = mid;
    }
    const uint64_t v = (uint64_t)raw[boff[lo] + hdr[lo] + (i - pbase[lo])];
    const uint32_t r = rank[i];
    if (rank[i + 1] != r) targets[r] = v;
    else positions[i - r] = v;
}

}  // namespace ffh
