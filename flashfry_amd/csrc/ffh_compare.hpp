// ffh_compare.hpp -- THE HOT KERNEL of the discover scan (gfx950, wave64): candidate guides x bucketed targets.
//
// Replaces the inner loops of BlockManager.compareBlock / compareLinearBlock / compareIndexedBlock (blocks/BlockManager.scala:63-254)
// and the bin traversals that feed them (OrderedBinTraversalFactory.scala:146-177, LinearTraversal.scala:64-97); the unit of work is
// BitEncoding.mismatches (bitcoding/BitEncoding.scala:127-132).  What reaches this kernel is, per scan image,
// a join of two streams that are both sorted by bucket: the targets of every bucket and the bucket's candidate guides (CSR).
//
// BIT-SLICED TARGETS.  A pair test done one pair per lane costs v_xor + v_bitop3 + v_bcnt + a share of a reduction = ~12.6 SIMD cycles
// per 64 pairs (v_bcnt, v_min*, v_cmp run at half rate on gfx950: tools/ubench/op_rate.hip), and that is what binds the scan.  Here a
// lane tests ONE candidate guide against THIRTY-TWO targets per step: the image stores the targets of a bucket in GROUPS of 32, a
// group being 2R + 1 words -- for each of the R bases the bucket id does not fix, bit t of word 2i / word 2i + 1 is the high / low
// plane bit of base i of target t, plus a word of valid bits.  With the guide's own plane bits spread into all-zero / all-one masks GH_i,
// GL_i (once per candidate),
//     mismatch_i = (H_i ^ GH_i) | (L_i ^ GL_i)             32 pairs in two full-rate instructions (v_xor, v_bitop3)
//     count      = carry-save adder tree over the R words  (v_bitop3 does a full adder's sum or carry in one instruction)
//     hit        = count <= maxMismatch - d  (& count > r1 on the suffix image) & valid        four to eight v_bitop3
// ~37 (R = 9) / ~46 (R = 11) full-rate instructions per 64 lanes x 32 pairs: ~1.2 instructions per 64 pairs instead of 3.75, none of
// them half rate, and no per-pair popcount at all.  d = the mismatches inside the bucket key, per (guide, bucket).
//
// WORK LAYOUT.  A wave owns a BATCH at a time: a run of consecutive small buckets (prefix image: ~72 targets = 3 groups per bucket) or a
// slice of one large bucket (suffix image: ~36 groups).  Both streams of a batch are contiguous in memory: the batch is fetched with a
// few wide coalesced loads one batch ahead of its use (bucket boundaries three, candidate ids two batches ahead), parked in the wave's
// LDS strip and computed out of LDS.  The batch's (candidate, group range) JOBS are dealt to the lanes, 64 to a row: a candidate of a
// small bucket is one job, a candidate of a large bucket is split into P jobs of ~6 groups, so a row's lanes run the same number of
// steps and every lane is busy whatever the bucket sizes are.  There are no per-bucket work items, no bucket tails, no sub-wave
// packing.  A lane with a hit (a non-zero 32-bit mask, rare) stages (guide, slot) per set bit; the stage leaves as sort keys
// (guide << tbits | database index) with one global atomic per ~250 hits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ffh_prims.hpp"

namespace ffh {

// The kernel's geometry.  (Same-box A/B builds of other values: tools/build_variant.sh rewrites its private copy of this block with
// `KW=768 STAGE=192 ...` arguments; the product has no build-time switches.)
constexpr int FFH_KW = 1024;            // group words parked per wave
constexpr int FFH_KC = 256;             // candidates parked per wave
constexpr int FFH_STAGE = 256;          // staged hits per wave
constexpr int FFH_WAVES_PER_SIMD = 4;   // launch bound (four blocks of four waves per CU: LDS-limited)
constexpr int FFH_GPL = 6;              // groups per job of a large bucket, about
constexpr int FFH_PIPE_TRIPS = 2;       // 16-byte pieces of a group's words requested one group ahead (more cost registers)
constexpr int FFH_MAX_ENTRY_WORK = 2048;   // group tests per work entry of a heavy bucket, at most
constexpr int FFH_ROW_SETUP_Q = 0;         // one-bucket pieces: quarters of a step a row's set-up is priced at when the parts per candidate are chosen.  0 = round 5's
                                           // price (rows x steps).  tools/lds_conflict_model.py: a row's set-up costs ~1.25 steps (60 against 48 instructions), and with
                                           // it priced the suffix image stops cutting a bucket into 12-13 parts of 3 groups -- two rows, and the parts p and p + 8 of
                                           // one candidate on the same LDS banks (a group there is an EVEN number of 16-byte pieces) -- in favour of 6 parts of 7 groups
                                           // in one row: predicted -2 % vector instructions, -11 % LDS cycles of the strip reads on that image.  A/B: FFH_ROW_SETUP_Q=5
                                           // (tools/build_variant.sh); unmeasured, hence 0
constexpr int FFH_TRIP_STATS = 0;          // 1 (tools/build_variant.sh only): the kernel counts its rows, steps, parks, pushes and flushes per image
                                           // (cursor[16..27], printed by scan_impl) -- the trip counts of profiles/r05/compare_attribution.md
constexpr int kCmpThreads = 256;           // four waves, each with its own LDS strip: no block-level synchronisation in the kernel
constexpr int kCmpWaves = kCmpThreads / 64;
constexpr int kKW = FFH_KW;                // group words parked per wave (4 KB: 51 groups of 20 words, 42 of 24)
constexpr int kKC = FFH_KC;                // candidates parked per wave
constexpr int kStage = FFH_STAGE;          // staged hits per wave
constexpr int kMaxNB = 15;                 // buckets per batch: lanes 0..15 (one DPP row) hold the batch's bucket boundaries
constexpr int kKeyRegs = kKW / 256;        // 16-byte loads per lane and batch
constexpr int kGidRegs = kKC / 64;
constexpr int kGroupsPerLane = FFH_GPL;    // a candidate of a large bucket is split into jobs of about this many groups
constexpr int kMaxParts = 16;             // jobs per candidate at most (park: fewer when a piece has many candidates)
constexpr int kMinRest = 7, kMaxRest = 12;   // rest-key widths k_compare has a row form for (19-mers with a 12-base bucket key: 7)

constexpr uint32_t kStatPairs = 4, kStatEntries = 6, kQueue = 64, kQueues = 16, kQueueStride = 16;   // cursor[64 + 16 (16 side + k)]: the side's k-th work queue (chunks drawn
// so far), every queue in a 128-byte line of its own (round 5: atomics on different words of ONE line queue behind each other like atomics on one word)
constexpr uint32_t kCounterWords = kQueue + 2 * kQueues * kQueueStride;   // size of a context's counter block
// cursor[4 + side]: executed pair tests, cursor[6 + side]: work entries of this launch
constexpr uint32_t kQueueChunkLong = 16, kQueueChunkMedium = 4;   // work entries a wave draws from its queue at a time (long lists: 8: 1.07, 16 / 32: 1.06 ms per launch;
                                                                  // the medium-length lists of a bounded scan's slabs: 4)

__host__ __device__ constexpr int group_words(int rest) { return (2 * rest + 1 + 3) & ~3; }   // words per group of 32 targets (16-byte multiple)

struct WorkEntry;
struct SideArgs {
    const uint32_t *gstart;   // [nb + 1] first group of every bucket
    const uint32_t *gwords;   // [groups * GW + kKW + 64] bit-sliced groups (padded: a batch is fetched in whole 16-byte pieces)
    const uint32_t *tidx;     // [groups * 32] database index of every slot (looked up when a hit leaves the wave); null for a direct image
    uint32_t dd_off;          // direct image (k_bucket_first): gstart[dd_off + b] = ddelta[b], database index of a slot = slot + ddelta[bucket]
                              // (the table sits behind gstart in one allocation: one base address for both loads); else 0
    const uint32_t *istart;   // [nb + 1] first candidate of every bucket (absolute index into gids)
    const uint2 *gtab;        // [guides of this batch] {rest key H << 16 | L, bucket id} of every guide on this side
    uint32_t nb;              // buckets
    uint32_t width;           // bases in the bucket id (= bits per plane of it)
    uint32_t rest;            // R: bases in the rest key
    uint32_t NB;              // buckets per batch
    uint32_t split;           // groups per work entry at most (= what the LDS strip holds)
    const WorkEntry *list;         // work entries: a batch that has candidates, or -- for a bucket larger than the strip (a repeat
                              // family) -- one strip-sized range of its groups, so that such a bucket is spread over many waves instead
                              // of pinning one
    const uint32_t *n_list;   // their number (device memory: built on the stream, no host round trip)
    uint32_t list_cap;        // entries the list holds (a guide set that needs more is noticed by the host after the launch, which
                              // then runs again with a larger list: scan_impl)
    int r_far;                // suffix side: r1 -- a pair is reported here only with MORE than r1 mismatches in its rest key (the
                              // prefix image reports the others); prefix side: -1
};
struct CompareArgs {
    SideArgs side[2];         // 0 prefix image, 1 suffix image
    const uint32_t *gids;     // candidate CSR of both sides
    uint64_t *hits;
    uint64_t cap;
    uint32_t guide_base[2];   // per side: first guide of this batch of guides in the caller's array
    int tbits;                // hit key = (global guide << tbits) | database index
    int max_mm;
    const uint32_t *gmap[2];  // per side, nullable: number, in the caller's guide array, of every guide of the side's candidate list (a
                              // bounded scan's later slabs run the suffix image on the packed set of guides still active, the prefix
                              // image on all guides with the retired ones made unreachable); null = guide_base + position
};

// before every compare launch: clears the per-launch statistics words -- 64 threads of the first k_guide_keys launch of the candidate lists
// (the first launch of the sequence the compare launch ends; a launch of its own, k_compare_setup, until round 5)
__device__ void compare_setup_words(unsigned long long *__restrict__ cursor, int first_batch, uint32_t t) {
    if (first_batch && t < 4) cursor[t] = 0ull;  // [0] hit cursor, [1] real hits: once per scan, they run across guide batches
    if (t >= 4 && t < 8) cursor[t] = 0ull;       // [4], [5] executed pairs, [6], [7] work entries of the two images: per launch
    if (t == 13) cursor[13] = 0ull;              // the list of heavy segments of the hit ordering (k_segsort)
    if (t < 2 * kQueues) cursor[kQueue + t * kQueueStride] = 0ull;  // the two images' work queues (k_compare)
    if (FFH_TRIP_STATS && t >= 16 && t < 30) cursor[t] = 0ull;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_c4;   // 32-bit LDS addresses

__device__ __forceinline__ void wave_lds_fence() {  // this wave's LDS writes are visible to its own later reads, in program order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_of(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// lane l gets lane l + 1's value / lane l - k's value inside its row of 16 lanes (0 past the row's end): DPP, no LDS round trip
__device__ __forceinline__ uint32_t row_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }   // row_shl:1
template <int K>
__device__ __forceinline__ uint32_t row_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + K, 0xf, 0xf, true); }  // row_shr:K

#define BITOP3(a, b, c, lut) __builtin_amdgcn_bitop3_b32((a), (b), (c), (lut))   // result bit = lut[(a << 2) | (b << 1) | c]

// number of set planes among n one-bit-per-target words: a carry-save adder tree, one v_bitop3 per sum / per carry
template <int N>
__device__ __forceinline__ void count_planes(const uint32_t (&m)[N], uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3) {
    static_assert(N >= 2 && N <= 15, "a four-bit count");
    uint32_t w[4][N];
    int n[4] = {N, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < N; ++i) w[0][i] = m[i];
    uint32_t c[4] = {0, 0, 0, 0};
#pragma unroll
    for (int lv = 0; lv < 4; ++lv) {
        int cnt = n[lv];
#pragma unroll
        for (int it = 0; it < N; ++it) {
            if (cnt >= 3) {   // full adder: three words of this weight -> one of this weight, one of the next
                const uint32_t a = w[lv][cnt - 1], b = w[lv][cnt - 2], d = w[lv][cnt - 3];
                cnt -= 3;
                w[lv][cnt++] = BITOP3(a, b, d, 0x96);
                if (lv < 3) w[lv + 1][n[lv + 1]++] = BITOP3(a, b, d, 0xE8);
            }
        }
        if (cnt == 2) {      // half adder
            const uint32_t a = w[lv][1], b = w[lv][0];
            w[lv][0] = a ^ b;
            cnt = 1;
            if (lv < 3) w[lv + 1][n[lv + 1]++] = a & b;
        }
        c[lv] = cnt ? w[lv][0] : 0u;
    }
    c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3];
}

// job -> bucket lookup of a parked piece: kMaxRows words of 64 marker bits, one bit per non-empty bucket at the job before its first
constexpr int kMaxRows = 16;               // rows (of 64 jobs) of one parked piece at most: park() caps the jobs of a piece at 1024

// what a wave keeps in LDS: one struct per wave, so that everything is an immediate offset from ONE base address (six separately
// declared arrays cost six scalar registers of bases, in a kernel that was spilling them)
struct alignas(16) WaveLds {
    uint32_t strip[kKW + 64];              // the batch's group words
    uint2 cand[kKC];                       // {rest key, bucket id} of the batch's candidates
    uint32_t gid[kKC];                     // their guide numbers
    uint4 tab[2][16];                      // per non-empty bucket of the piece: {first job, first candidate, bucket id, P | 65536 / P} / {strip word, groups, groups per job, first slot}
    uint64_t stage[kStage];                // staged hits
    uint64_t mark[kMaxRows];               // bucket markers
};

typedef __attribute__((address_space(3))) uint64_t lds_u64;
constexpr uint32_t kChunkMin = 512, kChunkMax = 8192;   // hit-buffer slots a wave reserves at a time from its second flush on: at least, at most
struct HitStage {
    WaveLds *W;              // W->stage: this wave's staged hits, guide of this batch << 32 | side << 31 | slot in the side's image
    uint32_t fill;
    uint32_t lane;
    const CompareArgs *A;    // the kernel's argument block (scalar loads from the kernarg segment at flush time)
    unsigned long long *cursor;
    // The wave owns a CHUNK of the hit buffer and fills it flush by flush; only a new chunk costs an atomic on the one global cursor
    // (same-address atomics complete at ~90 per microsecond on this part: with one per flush a 5-mismatch or repeat-rich scan spent
    // most of its time queueing there).  Chunks grow with the wave's own hit count, so a wave with few hits wastes little and a wave
    // with many asks rarely; what is left of the last chunk is filled with all-ones keys, which sort behind every hit.
    unsigned long long chunk_pos;
    uint32_t chunk_left;
    uint32_t n_real;         // (a scan of 2^32 raw hits or more is refused by the host)
    uint32_t st_push = 0, st_flush = 0, st_flush_it = 0;   // FFH_TRIP_STATS

    __device__ __forceinline__ void flush() {
        wave_lds_fence();
        lds_u64 *my = (lds_u64 *)W->stage;
        uint64_t *__restrict__ hits = A->hits;
        const uint64_t cap = A->cap;
        const uint32_t *__restrict__ tidx_p = A->side[0].tidx, *__restrict__ tidx_s = A->side[1].tidx;
        const uint32_t base_p = A->guide_base[0], base_s = A->guide_base[1];
        const uint32_t *__restrict__ gmap_p = A->gmap[0], *__restrict__ gmap_s = A->gmap[1];
        const int tbits = A->tbits;
        const unsigned long long old_pos = chunk_pos;
        const uint32_t old_left = chunk_left;
        unsigned long long new_pos = 0;
        if (fill > old_left) {
            // Every reservation is an atomic on the ONE hit cursor, and same-address atomics complete at ~90 per microsecond on this
            // part: at hg38 scale (4096 waves x ~2800 hits, a flush per ~190) reserving exactly what a flush holds kept the cursor busy
            // two thirds of the launch (1.015 -> 0.945 ms with chunks of 512; the repeat-structured workload's compare 4.1 -> 2.7 ms:
            // profiles/r04/ab_log.txt 15).  So: a wave's FIRST flush reserves exactly what it holds (a wave of a small scan flushes once,
            // at its end: no padding, the hit buffer of a chr22-scale call stays a few thousand keys), every further one at least
            // kChunkMin slots, and past 4096 hits an eighth of what the wave has produced so far (at most kChunkMax): the padding left
            // at the end stays a few per cent of the hits, the number of atomics per wave small.
            const uint32_t need = fill - old_left;
            const uint32_t want = n_real == 0u ? need : n_real < 4096u ? max(need, kChunkMin) : max(min(n_real >> 3, kChunkMax), need);
            if (lane == 0) new_pos = atomicAdd(cursor, (unsigned long long)want);
            new_pos = ((unsigned long long)uni((uint32_t)(new_pos >> 32)) << 32) | uni((uint32_t)new_pos);
            chunk_pos = new_pos + (fill - old_left);
            chunk_left = want - (fill - old_left);
        } else {
            chunk_pos = old_pos + fill;
            chunk_left = old_left - fill;
        }
        // a record leaves as the sort key (global guide << tbits) | database index -- the index lookup rides on the flush instead of
        // a pass of its own over all hits.  (Requesting the look-ups of all staged records -- four per lane -- before the first key
        // is stored, for the rows of a repeat family where a wave flushes after every third step: nothing gained there, 1.12
        // against 1.04 ms per launch at hg38 scale, where a flush holds a handful of records; dropped.)
        if (FFH_TRIP_STATS) { ++st_flush; st_flush_it += (fill + 63u) >> 6; }
        for (uint32_t i = lane; i < fill; i += 64) {
            const unsigned long long dst = i < old_left ? old_pos + i : new_pos + (i - old_left);
            if (dst < cap) {
                const uint64_t h = my[i];
                const uint32_t lo = (uint32_t)h, slot = lo & 0x7FFFFFFFu;
                // (a direct image's records already hold the database index: no lookup, no 128-byte line per hit)
                const bool sfx = (lo >> 31) != 0u;
                uint32_t ti = slot;
                if (sfx && tidx_s) ti = tidx_s[slot];
                if (!sfx && tidx_p) ti = tidx_p[slot];
                const uint32_t gl = (uint32_t)(h >> 32);
                uint32_t g = gl + (sfx ? base_s : base_p);
                if (sfx && gmap_s) g = gmap_s[gl];
                if (!sfx && gmap_p) g = gmap_p[gl];
                hits[dst] = ((uint64_t)g << tbits) | ti;
            }
        }
        wave_lds_fence();
        n_real += fill;
        fill = 0;
    }
    // end of the kernel: the rest of the stage, the padding of the last chunk, the wave's count of real hits
    __device__ __forceinline__ void finish() {
        if (fill) flush();
        uint64_t *__restrict__ hits = A->hits;
        const uint64_t cap = A->cap;
        for (uint32_t i = lane; i < chunk_left; i += 64)
            if (chunk_pos + i < cap) hits[chunk_pos + i] = ~0ull;
        if (lane == 0 && n_real) atomicAdd(cursor + 1, (unsigned long long)n_real);
    }
    // wave-uniform call: `lanes` = ballot of `hit`
    __device__ __forceinline__ void push(uint64_t lanes, bool hit, uint32_t gid, uint32_t slot) {
        if (hit) ((lds_u64 *)W->stage)[fill + mbcnt(lanes)] = ((uint64_t)gid << 32) | slot;
        if (FFH_TRIP_STATS) ++st_push;
        fill += (uint32_t)__popcll(lanes);
        if (fill > kStage - 64) flush();  // always leave room for one more wave-wide batch
    }
};

// what a row's lanes share
struct RowCtx {
    HitStage *hs;
    int max_mm, r_far;
    uint32_t width;
    uint32_t st_rows = 0, st_steps = 0, st_hit_steps = 0, st_lane_steps = 0;   // FFH_TRIP_STATS (per image: reset by run_side)
};

// One row: 64 jobs, one per lane -- one candidate guide against `trips` consecutive groups of its bucket.
//   rest: the guide's rest key (H << 16 | L), d: its mismatches inside the bucket key, gid: its id; strip: the wave's parked group
//   words; gword: strip word of the lane's first group; sbase: slot number of that group's first target (a direct image: its
//   database index)
//   FAR: what is known at compile time about the suffix image's extra condition "more than r1 mismatches in the rest key":
//   0 none (prefix image), 1 .. 4 count >= FAR as one or two v_bitop3 on the count's bit planes, -1 taken from c.r_far at run time
//   EARLY: 16-byte pieces of a group's words requested one group ahead (registers: the any-width instance cannot afford them)
template <int R, int FAR, int EARLY, int SIDE>
__device__ __forceinline__ void scan_row(RowCtx &c, uint32_t rest, uint32_t d, uint32_t gid, uint32_t trips, uint32_t gword, uint32_t sbase,
                                         const uint32_t *strip) {
    constexpr int GW = group_words(R);
    lds_c4 *gp = (lds_c4 *)(strip + gword);            // per lane, 16-byte aligned (GW is a multiple of 4)
    uint32_t w[GW];
    // the group's words come in GW / 4 16-byte LDS reads; the first kEarly of them are requested one group ahead (below)
    constexpr int kEarly = EARLY < GW / 4 ? EARLY : GW / 4;
    auto fetch = [&](auto from, auto to) {
#pragma unroll
        for (int q = decltype(from)::value; q < decltype(to)::value; ++q) {
            const u32x4 v = gp[q];
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using IE = std::integral_constant<int, kEarly>;
    using IN = std::integral_constant<int, GW / 4>;
    fetch(I0{}, IE{});                                 // the first group is requested before the masks are set up
    // the guide's plane bits as masks, its mismatch budget inside this bucket as threshold masks
    uint32_t GH[R], GL[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        GH[i] = (uint32_t)__builtin_amdgcn_sbfe((int)rest, 16 + i, 1);   // 0 or ~0
        GL[i] = (uint32_t)__builtin_amdgcn_sbfe((int)rest, i, 1);
    }
    const int thr = c.max_mm - (int)d;                 // rest mismatches allowed
    if (thr < 0) trips = 0;
    const uint32_t tcl = (uint32_t)min(max(thr, 0), 15);
    const uint32_t t0 = 0u - (tcl & 1u), t1 = 0u - ((tcl >> 1) & 1u), t2 = 0u - ((tcl >> 2) & 1u), t3 = 0u - ((tcl >> 3) & 1u);
    const uint32_t far = (uint32_t)(c.r_far + 1);      // suffix image: at least this many rest mismatches (0: no condition)
    const uint32_t f0 = 0u - (far & 1u), f1 = 0u - ((far >> 1) & 1u), f2 = 0u - ((far >> 2) & 1u), f3 = 0u - ((far >> 3) & 1u);
    // The loop is wave-uniform (the hit stage is the wave's): `live` = the lanes whose job still has a group at step t.  It is both the
    // loop condition and -- as the mask operand of one v_cndmask -- what silences a lane that is done (its reads run into a neighbour's
    // groups): one compare per step for both.  (Written as `for (; ballot(t < trips);) ... if (t >= trips) hit = 0` the compiler
    // materialised the condition in a register and compared twice: five vector instructions per step instead of two.)
    uint64_t live = __builtin_amdgcn_ballot_w64(trips != 0u);
    if (FFH_TRIP_STATS) ++c.st_rows;
    for (uint32_t t = 0; live; ++t) {
        if (FFH_TRIP_STATS) { ++c.st_steps; c.st_lane_steps += (uint32_t)__popcll(live); }
        fetch(IE{}, IN{});
        uint32_t m[R];
#pragma unroll
        for (int i = 0; i < R; ++i) m[i] = BITOP3(w[2 * i + 1], GL[i], w[2 * i] ^ GH[i], 0xBE);   // (L ^ GL) | (H ^ GH): base i differs
        const uint32_t valid = w[2 * R];
        gp += GW / 4;
        // The first words of the NEXT group are requested here: w[] is dead from this point on, so their LDS round trip runs under the
        // adder tree and the hit handling of this group, and the rest of the group is requested while they are being used.  (After its
        // last group a lane reads the group behind it -- a neighbour's, or the strip's padding -- and drops it.)
        fetch(I0{}, IE{});
        uint32_t c0, c1, c2, c3;
        count_planes<R>(m, c0, c1, c2, c3);
        // count <= thr, bit-serially from the low end: e_k = "the low k + 1 bits of count <= those of thr"
        uint32_t e = BITOP3(c0, t0, t0, 0xCF);          // ~c0 | t0
        e = BITOP3(c1, t1, e, 0x8E);                    // (~c & t) | (~(c ^ t) & e)
        e = BITOP3(c2, t2, e, 0x8E);
        e = BITOP3(c3, t3, e, 0x8E);
        uint32_t hit = e & valid;
        if constexpr (FAR == 1) hit = BITOP3(BITOP3(c0, c1, c2, 0xFE), c3, hit, 0xA8);         // count >= 1: (c0 | c1 | c2 | c3) & hit
        else if constexpr (FAR == 2) hit &= BITOP3(c1, c2, c3, 0xFE);                          // count >= 2: c1 | c2 | c3
        else if constexpr (FAR == 3) hit = BITOP3(BITOP3(c0, c1, c2, 0xEA), c3, hit, 0xA8);    // count >= 3: ((c0 & c1) | c2 | c3) & hit
        else if constexpr (FAR == 4) hit = BITOP3(c2, c3, hit, 0xA8);                          // count >= 4: (c2 | c3) & hit
        else if constexpr (FAR < 0) {
            if (c.r_far >= 0) {                         // uniform: count >= far, bit-serially like the threshold
                uint32_t g = BITOP3(c0, f0, f0, 0xF3);  // c0 | ~f0
                g = BITOP3(c1, f1, g, 0xB2);            // (c & ~f) | (~(c ^ f) & g)
                g = BITOP3(c2, f2, g, 0xB2);
                g = BITOP3(c3, f3, g, 0xB2);
                hit &= g;
            }
        }
        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(hit) : "v"(hit), "s"(live));   // a lane that is done
        live = __builtin_amdgcn_ballot_w64(t + 1u < trips);
        const uint64_t lanes = __builtin_amdgcn_ballot_w64(hit != 0u);
        if (lanes) {   // rare per lane, usual per wave: the lanes with a non-zero mask stage one record per set bit (almost always one)
            if (FFH_TRIP_STATS) ++c.st_hit_steps;
            const uint32_t slot0 = (sbase + (t << 5)) | ((uint32_t)SIDE << 31);
            uint64_t more = lanes;
            do {
                c.hs->push(more, hit != 0u, gid, slot0 + (uint32_t)__ffs((int)hit) - 1u);
                hit &= hit - 1u;
                more = __builtin_amdgcn_ballot_w64(hit != 0u);
            } while (more);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The work list of one image for one compare launch, built from the bucket boundaries and the candidate CSR offsets:
//   k_work_fill   per batch of NB consecutive buckets: no entry if it has no candidate or no target, else ceil(groups / split) x
//                 ceil(candidates / kKC) of them -- an entry never holds more than the wave's strip and candidate table take (a repeat
//                 family's bucket with thousands of candidate guides used to be ONE entry per 51 groups, worked off piece by piece by
//                 one wave while the launch waited for it).  A call with a handful of guides lists a few hundred batches instead of making every wave walk all
//                 4^11 buckets' boundaries; a bucket of 1e5 targets becomes ~100 entries dealt to ~100 waves.
// ---------------------------------------------------------------------------------------------------------
// One slab of a bounded scan on the prefix image, whose candidate CSR is built ONCE for all slabs: the batch's buckets that belong to
// the slab (first three bases ranked lo .. hi in sequence order: bucket_rank).  The rank changes every 4^(width - 3) bucket ids, a
// batch has at most kMaxNB of them, so the slab's part of a batch is one run [s0, s1).  lo = 0, hi = 63: the whole batch.
__device__ __forceinline__ void slab_run(uint32_t b0, uint32_t b1, uint32_t lo, uint32_t hi, uint32_t width, uint32_t &s0, uint32_t &s1) {
    s0 = b0; s1 = b1;
    if (lo == 0u && hi >= 63u) return;
    while (s0 < b1) { const uint32_t r = bucket_rank(s0, width); if (r >= lo && r <= hi) break; ++s0; }
    s1 = s0;
    while (s1 < b1) { const uint32_t r = bucket_rank(s1, width); if (r < lo || r > hi) break; ++s1; }
}
// How a batch is cut into work entries: n_g pieces of at most `split` groups x n_c chunks of at most kKC >> cs candidates.  A batch
// of ordinary buckets is one piece and -- unless a skewed guide set piles candidates on it -- one chunk.  A bucket larger than the
// strip (n_g > 1) or a batch of one bucket (the suffix image at genome scale) is a repeat family's when it also has many candidates,
// and its rows are then full of hits, each costing more than the test that found it: one entry of 42 groups x 256 candidates kept a
// wave busy for a millisecond while the launch waited for it.  Such a batch takes smaller candidate chunks, so that an entry stays
// within kMaxEntryWork group tests.
constexpr uint32_t kMaxEntryWork = FFH_MAX_ENTRY_WORK, kMaxCandShift = 4;
struct WorkSplit { uint32_t n_g, n_c, cs; };
__device__ __forceinline__ WorkSplit work_split(uint32_t ngr, uint32_t nc, uint32_t split, bool one_bucket) {
    WorkSplit w;
    w.n_g = (ngr + split - 1u) / split;
    w.cs = 0;
    const uint32_t piece = min(ngr, split);
    if (w.n_g > 1u || one_bucket)
        while (w.cs < kMaxCandShift && ((uint32_t)kKC >> w.cs) * piece > kMaxEntryWork) ++w.cs;
    const uint32_t cs_n = (uint32_t)kKC >> w.cs;
    w.n_c = (nc + cs_n - 1u) / cs_n;
    return w;
}
// (Round 5: two launches per image instead of four.  k_work_count leaves, besides every batch's number of entries, the sum of each block
// of 1024 batches; a block of k_work_fill adds up the sums of the blocks before it (a few hundred values) and scans its own 1024 counts:
// the device-wide scan between the two -- two launches -- is gone and the list keeps its bucket order, which the compare launch is
// sensitive to: handing the blocks their ranges through an atomic, in order of arrival, cost 5 % of the compare launch at hg38 scale
// and 40 % on the repeat-structured workload, profiles/r05/ab_log.txt 5.)
constexpr int kWorkThreads = 1024;
// (both images' lists in one launch: blockIdx.y picks the image, see k_guide_keys)
struct WorkEntry;
struct WorkArgs {
    const uint32_t *gstart, *istart; uint32_t nb, NB, split, n_bat; uint32_t *counts, *block_sums;
    const unsigned long long *part_pairs; uint32_t n_part; unsigned long long *pairs_out;
    WorkEntry *list; uint32_t list_cap; unsigned long long *n_out; uint32_t *n_list /* the same number where the compare launch reads it: a line of its own */;
    uint32_t rank_lo, rank_hi, width, grid;
};
__global__ __launch_bounds__(kWorkThreads) void k_work_count(WorkArgs a0, WorkArgs a1) {
    const WorkArgs &A = blockIdx.y ? a1 : a0;
    if (blockIdx.x >= A.grid) return;
    const uint32_t *__restrict__ gstart = A.gstart, *__restrict__ istart = A.istart;
    uint32_t *__restrict__ counts = A.counts, *__restrict__ block_sums = A.block_sums;
    const unsigned long long *__restrict__ part_pairs = A.part_pairs;
    unsigned long long *__restrict__ pairs_out = A.pairs_out;
    const uint32_t nb = A.nb, NB = A.NB, split = A.split, n_bat = A.n_bat, n_part = A.n_part, rank_lo = A.rank_lo, rank_hi = A.rank_hi, width = A.width;
    __shared__ uint32_t wsum[kWorkThreads / 64];
    __shared__ unsigned long long red[kWorkThreads / 64];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    uint32_t n = 0;
    if (t < n_bat) {
        uint32_t s0, s1;
        slab_run(t * NB, min(nb, t * NB + NB), rank_lo, rank_hi, width, s0, s1);
        const uint32_t ngr = gstart[s1] - gstart[s0], nc = istart[s1] - istart[s0];
        const WorkSplit w = work_split(ngr, nc, split, s1 - s0 == 1u);
        n = (ngr && nc) ? w.n_g * w.n_c : 0u;
        counts[t] = n;
    }
    uint32_t sum = n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0) wsum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < kWorkThreads / 64; ++k) tot += wsum[k];
        block_sums[blockIdx.x] = tot;
    }
    if (blockIdx.x != 0 || !n_part) return;
    // the executed-pair statistic (targets x candidates of every bucket): k_item_bin left one partial sum per partition
    unsigned long long pairs = 0;
    for (uint32_t d = threadIdx.x; d < n_part; d += blockDim.x) pairs += part_pairs[d];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) pairs += __shfl_xor(pairs, d, 64);
    if (lane == 0) red[threadIdx.x >> 6] = pairs;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int k = 0; k < kWorkThreads / 64; ++k) tot += red[k];
        atomicAdd(pairs_out, tot);
    }
}
// A work entry, 32 bytes: what a wave needs to fetch the batch -- first bucket, buckets, groups [g0, g1), candidates [c0, c1) (absolute
// indices into gwords / gids) -- worked out HERE, once, from the bucket boundaries.  (Round 3's 16-byte entries held {first bucket,
// buckets | chunk, first group, end group} and every wave derived the extents from the boundaries with a dozen v_readlane + scalar
// min / max per batch and pipeline stage; the boundaries had to be in flight three batches ahead for it.)
struct WorkEntry { uint32_t b0, nbv, g0, g1, c0, c1, pad0, pad1; };
static_assert(sizeof(WorkEntry) == 32, "two 16-byte stores");
// A thread per batch; a batch with many entries (a repeat family's bucket: hundreds) is written by its whole wave.
__global__ __launch_bounds__(kWorkThreads) void k_work_fill(WorkArgs a0, WorkArgs a1) {
    const WorkArgs &A = blockIdx.y ? a1 : a0;
    if (blockIdx.x >= A.grid) return;
    const uint32_t *__restrict__ gstart = A.gstart, *__restrict__ istart = A.istart, *__restrict__ counts = A.counts, *__restrict__ block_sums = A.block_sums;
    WorkEntry *__restrict__ list = A.list;
    unsigned long long *__restrict__ n_out = A.n_out;
    uint32_t *__restrict__ n_list = A.n_list;
    const uint32_t nb = A.nb, NB = A.NB, split = A.split, n_bat = A.n_bat, list_cap = A.list_cap, rank_lo = A.rank_lo, rank_hi = A.rank_hi, width = A.width;
    __shared__ uint32_t scan_lds[16];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    uint32_t o = 0, n = t < n_bat ? counts[t] : 0u, s0 = 0, s1 = 0, gs = 0, ge = 0, is0 = 0, is1 = 0;
    {   // entries before this block (the sums of the blocks before it) + before this batch inside the block
        uint32_t before = 0;
        for (uint32_t b = threadIdx.x; b < blockIdx.x; b += kWorkThreads) before += block_sums[b];
        uint32_t incl = n, binc = before;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += v;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) binc += __shfl_xor(binc, d, 64);
        if (lane == 63) scan_lds[threadIdx.x >> 6] = incl;
        __shared__ uint32_t before_lds[kWorkThreads / 64];
        if (lane == 0) before_lds[threadIdx.x >> 6] = binc;
        __syncthreads();
        uint32_t wave_off = 0, tot = 0, base = 0;
#pragma unroll
        for (int k = 0; k < kWorkThreads / 64; ++k) {
            const uint32_t v = scan_lds[k];
            if ((uint32_t)k < (threadIdx.x >> 6)) wave_off += v;
            tot += v;
            base += before_lds[k];
        }
        o = base + wave_off + incl - n;
        if (blockIdx.x == A.grid - 1 && threadIdx.x == 0) { *n_out = base + tot; *n_list = base + tot; }
    }
    WorkSplit w{1u, 1u, 0u};
    if (n) {
        slab_run(t * NB, min(nb, t * NB + NB), rank_lo, rank_hi, width, s0, s1);
        gs = gstart[s0]; ge = gstart[s1];
        is0 = istart[s0]; is1 = istart[s1];
        w = work_split(ge - gs, is1 - is0, split, s1 - s0 == 1u);
    }
    // entry k of the batch: piece kg of its groups x chunk kc of its candidates (kKC >> cs each)
    auto entry = [&](uint32_t k, uint32_t o_, uint32_t s0_, uint32_t nbv_, uint32_t gs_, uint32_t ge_, uint32_t is0_, uint32_t is1_, uint32_t n_g_, uint32_t cs_) {
        const uint32_t kc = k / n_g_, kg = k - kc * n_g_;   // (the pieces of one candidate chunk are neighbours; the queue deals neighbours to different waves)
        if (o_ + k >= list_cap) return;
        const uint32_t cs_n = (uint32_t)kKC >> cs_;
        uint32_t g0 = gs_ + kg * split, g1 = min(ge_, gs_ + (kg + 1u) * split), c0 = is0_ + kc * cs_n, c1 = min(is1_, c0 + cs_n);
        if (g1 <= g0) { g1 = g0; c1 = c0; }   // nothing to compare the candidates with
        uint4 *dst = reinterpret_cast<uint4 *>(list + (o_ + k));
        dst[0] = make_uint4(s0_, nbv_, g0, g1);
        dst[1] = make_uint4(c0, c1, 0u, 0u);
    };
    constexpr uint32_t kSerial = 4;
    if (n && n <= kSerial)
        for (uint32_t k = 0; k < n; ++k) entry(k, o, s0, s1 - s0, gs, ge, is0, is1, w.n_g, w.cs);
    for (uint64_t big = __ballot(n > kSerial); big; big &= big - 1ull) {
        const int src = __ffsll((long long)big) - 1;
        const uint32_t n_b = __shfl(n, src, 64), o_b = __shfl(o, src, 64), s0_b = __shfl(s0, src, 64), nbv_b = __shfl(s1 - s0, src, 64);
        const uint32_t gs_b = __shfl(gs, src, 64), ge_b = __shfl(ge, src, 64), ng_b = __shfl(w.n_g, src, 64), cs_b = __shfl(w.cs, src, 64);
        const uint32_t is0_b = __shfl(is0, src, 64), is1_b = __shfl(is1, src, 64);
        for (uint32_t k = lane; k < n_b; k += 64) entry(k, o_b, s0_b, nbv_b, gs_b, ge_b, is0_b, is1_b, ng_b, cs_b);
    }
}

// R0 / R1: the rest widths of the prefix / suffix image this instance is compiled for (the row forms are fixed and the kernel's
// registers are those of max(R0, R1)).  FAR1: r1 + 1 of the suffix image when the instance is compiled for it (scan_row), -1 = read
// from the arguments.  The host picks the instance (launch_compare).
// QC: the work entries are dealt through queues in chunks of QC (16 for long lists, 4 for medium ones); 0: a fixed stride (the host
// decides from the list lengths it expects: launch_compare)
template <int R0, int R1, int FAR1, int QC>
__global__ __launch_bounds__(kCmpThreads, FFH_WAVES_PER_SIMD) void k_compare(const CompareArgs A, unsigned long long *__restrict__ cursor) {
    __shared__ WaveLds lds_all[kCmpWaves];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = uni(threadIdx.x >> 6);
    WaveLds &W = lds_all[wave];
    HitStage hs{&W, 0u, lane, &A, cursor, 0ull, 0u, 0u};
    RowCtx rc{&hs, min(max(A.max_mm, 0), 30), -1, 0u};

    // The suffix image's work entries come first (they are the heavier ones), then the prefix image's; inside a side the wave runs a
    // software pipeline over the entries it takes.  The side is invariant in the pipeline, so everything that describes it stays in
    // scalar registers -- and only what describes THIS side: the other side's pointers are not read before its turn.
    auto run_side = [&](auto side_tag) {
        constexpr int side = decltype(side_tag)::value;
        constexpr int RC = side ? R1 : R0;             // compiled rest width of this side (0: any)
        constexpr int FARC = side ? FAR1 : 0;          // the prefix image has no far condition
        const SideArgs &S = A.side[side];
        if (!S.n_list) return;
        const uint32_t n_total = min(*S.n_list, S.list_cap);
        // The order in which a wave takes work entries.  Entries differ in weight (buckets differ in size, a repeat family's rows are
        // full of hits), and with a fixed stride the launch ended when its unluckiest wave did: 22 % after the mean wave at hg38 scale,
        // 2 .. 6 x the mean on the repeat-structured workload.  So the entries are dealt in chunks of QC: chunk t stands for
        // the entries t, t + C, t + 2 C, ... (C = number of chunks: the pieces of one heavy bucket, neighbours in the list, go to
        // different waves), a wave's first chunk is its own number, every further one the next ticket of its queue -- one atomic with
        // its round trip per QC entries, drawn where the wave has nothing in flight that it could not wait for.  kQueues
        // queues per side, each with its share of the waves and of the chunks: same-address atomics complete at ~90 per microsecond
        // on this part, and one queue for 4096 waves made the draws themselves the bottleneck (chunks of 4: 1.98 ms per launch
        // against 1.14 with the fixed stride).
        // kEnd: no chunk left (tickets only grow, so it stays that way); kNone: a chunk's entry past the end of the list.
        constexpr uint32_t kEnd = 0xFFFFFFFFu, kNone = 0xFFFFFFFEu;
        // (a list with fewer than two chunks per wave -- a small database, a bin shard, the slabs of a bounded scan -- keeps the fixed
        // stride: chunks of 16 would leave most waves without work, and smaller chunks mean a draw, with its exposed round trip,
        // per entry or two: measured 0.46 against 0.29 ms per step at chr22 scale, 0.39 against 0.25 ms per launch on an eighth
        // of hg38.  Lists in between -- the slabs of a bounded scan on a genome-scale database -- take chunks of 4 (9.24 against 9.48 ms
        // per step of the repeat-structured workload).  The host picks the instance from the list lengths it expects; decided in here,
        // per side at run time, the queue's gain at hg38 scale was gone -- 1.13 against 1.05 ms per launch.)
        constexpr uint32_t chunk_len = QC ? (uint32_t)QC : 1u;
        const uint32_t n_waves = gridDim.x * kCmpWaves;
        const uint32_t n_chunks = (n_total + chunk_len - 1u) / chunk_len;
        uint32_t chunk = blockIdx.x * kCmpWaves + wave, chunk_j = 0, stride_q = chunk;
        const uint32_t n_queues = min(kQueues, gridDim.x), my_queue = blockIdx.x % n_queues;   // (every queue has a block that draws from it)
        auto next_q = [&]() -> uint32_t {   // uniform
            if constexpr (QC == 0) { const uint32_t r = stride_q < n_total ? stride_q : kEnd; stride_q += n_waves; return r; }
            if (chunk_j == chunk_len) {
                uint32_t t = 0;
                if (chunk < n_chunks && lane == 0) t = (uint32_t)atomicAdd(cursor + kQueue + (side * kQueues + my_queue) * kQueueStride, 1ull);
                chunk = chunk < n_chunks ? n_waves + my_queue + n_queues * uni(t) : chunk;
                chunk_j = 0;
            }
            if (chunk >= n_chunks) return kEnd;
            const uint32_t r = chunk + chunk_j * n_chunks;
            ++chunk_j;
            return r < n_total ? r : kNone;
        };
        const uint32_t *__restrict__ gstart = S.gstart, *__restrict__ istart = S.istart, *__restrict__ gwords = S.gwords;
        const uint32_t *__restrict__ list = reinterpret_cast<const uint32_t *>(S.list);
        const uint32_t *__restrict__ gids = A.gids;
        const uint2 *__restrict__ gtab = S.gtab;
        const uint32_t dd_off = S.dd_off;
        constexpr uint32_t GW = (uint32_t)group_words(RC);
        rc.r_far = S.r_far;
        rc.width = S.width;
        uint32_t st_parks = 0;
        const uint32_t push0 = hs.st_push;
        rc.st_rows = rc.st_steps = rc.st_hit_steps = rc.st_lane_steps = 0;
        auto stats_out = [&]() {
            if (FFH_TRIP_STATS && lane == 0) {
                atomicAdd(cursor + 16 + side * 6 + 0, (unsigned long long)rc.st_rows); atomicAdd(cursor + 16 + side * 6 + 1, (unsigned long long)rc.st_steps);
                atomicAdd(cursor + 16 + side * 6 + 2, (unsigned long long)st_parks); atomicAdd(cursor + 16 + side * 6 + 3, (unsigned long long)(hs.st_push - push0));
                atomicAdd(cursor + 16 + side * 6 + 4, (unsigned long long)rc.st_hit_steps); atomicAdd(cursor + 16 + side * 6 + 5, (unsigned long long)rc.st_lane_steps);
            }
        };

        // ---- the pipeline.  A work entry lives in lanes 0..5 of ONE vector register ({first bucket, buckets, g0, g1, c0, c1}: scalar
        //      registers are the scarce resource here) and is read out with v_readlane where a stage needs it.  An entry past the end
        //      of the list, or of a chunk, is all zero: no groups, no candidates, its loads are skipped. ----
        auto load_entry = [&](uint32_t qq) -> uint32_t {
            uint32_t v = 0;
            if (qq < n_total && lane < 6) v = list[(size_t)qq * 8 + lane];
            return v;
        };
        // bucket boundaries of the entry: lane l <= nbv holds gstart / istart of bucket b0 + l; on a direct image lane 16 + l holds
        // ddelta of that bucket in dG (the second DPP row of the same register: no register of its own in the pipeline)
        auto load_desc = [&](uint32_t E, uint32_t &dG, uint32_t &dI) {
            dG = 0; dI = 0;
            const uint32_t nbv = lane_of(E, 1);
            if (nbv) {
                const uint32_t idx = lane_of(E, 0) + min(lane & 15u, nbv);
                dG = gstart[idx + ((lane & 48u) == 16u ? dd_off : 0u)];
                dI = istart[idx];
            }
        };
        // groups [g0, g1) of the image: g * GW words each, contiguous, fetched in 16-byte pieces (the array is padded by kKW + 64 words)
        auto load_groups = [&](uint32_t g0, uint32_t g1, uint4 (&kreg)[kKeyRegs]) {
            const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(gwords + (size_t)g0 * GW);
            const uint32_t nw = (g1 - g0) * GW;
#pragma unroll
            for (int j = 0; j < kKeyRegs; ++j)
                if ((uint32_t)j * 256u < nw) kreg[j] = src[j * 64 + lane];   // uniform predicate
        };
        auto load_gids = [&](uint32_t c0, uint32_t c1, uint32_t (&greg)[kGidRegs]) {
#pragma unroll
            for (int j = 0; j < kGidRegs; ++j) {
                const uint32_t c = c0 + (uint32_t)j * 64u + lane;
                greg[j] = 0;
                if (c < c1) greg[j] = gids[c];
            }
        };
        auto load_entries = [&](uint32_t c0, uint32_t c1, const uint32_t (&greg)[kGidRegs], uint2 (&ereg)[kGidRegs]) {
#pragma unroll
            for (int j = 0; j < kGidRegs; ++j) {
                const uint32_t c = c0 + (uint32_t)j * 64u + lane;
                if (c < c1) ereg[j] = gtab[greg[j]];
            }
        };

        // parks a piece of the batch -- groups [g0, g1), candidates [c0, c1) -- in the wave's LDS strip and deals its jobs; returns the
        // number of jobs.  Lane i < nbv: bucket i of the batch (dG / dI: its first group / first candidate, all buckets of the batch)
        auto park = [&](uint32_t b0, uint32_t nbv, uint32_t g0, uint32_t g1, uint32_t c0, uint32_t c1, const uint4 (&kreg)[kKeyRegs], const uint32_t (&greg)[kGidRegs],
                        const uint2 (&ereg)[kGidRegs], uint32_t dG, uint32_t dI) -> uint32_t {
            wave_lds_fence();  // the previous piece's reads are done
            const uint32_t nw = (g1 - g0) * GW;
#pragma unroll
            for (int j = 0; j < kKeyRegs; ++j)
                if ((uint32_t)j * 256u < nw) reinterpret_cast<uint4 *>(W.strip)[j * 64 + lane] = kreg[j];
#pragma unroll
            for (int j = 0; j < kGidRegs; ++j) {
                const uint32_t c = c0 + (uint32_t)j * 64u + lane;
                if (c < c1) {
                    W.cand[(uint32_t)j * 64u + lane] = ereg[j];
                    W.gid[(uint32_t)j * 64u + lane] = greg[j];
                }
            }
            if (lane < (uint32_t)kMaxRows) W.mark[lane] = 0ull;
            // bucket i: its groups and candidates clipped to the piece; P jobs per candidate
            const uint32_t nG = row_next(dG), nI = row_next(dI);
            const uint32_t lo_g = min(max(dG, g0), g1), hi_g = min(max(nG, g0), g1);
            const uint32_t lo_c = min(max(dI, c0), c1), hi_c = min(max(nI, c0), c1);
            const bool act = lane < nbv;
            const uint32_t ngr = act ? hi_g - lo_g : 0u, ng = (act && ngr) ? hi_c - lo_c : 0u;
            // Jobs per candidate: P parts of `per` groups each.  At most kMaxParts, and few enough that the piece stays within kMaxRows
            // rows of jobs.  (Divisions by a per-lane P <= 16 of numbers < 4096 go through v_rcp_f32: the quotients' fractional parts
            // are multiples of 1/P >= 1/16, far above the reciprocal's error, so a small bias makes floor / ceil exact; the integer
            // division the compiler emits instead costs ~30 instructions, and there were three per batch.)
            const uint32_t ncand = c1 - c0, pmax = ncand <= 64u ? 16u : ncand <= 128u ? 8u : 4u;
            auto ceil_div = [](uint32_t a, uint32_t b) { return (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)b) + 0.99f); };   // a < 4096, 1 <= b <= 16
            uint32_t P, per;
            if (nbv == 1u) {
                // One bucket in the piece (the suffix image, a large prefix bucket): lanes 0..15 price P = lane + 1 -- rows of 64 jobs
                // x groups per job -- and the cheapest wins.  (~ngr / 6 regardless of the candidates left every second piece of the
                // suffix image with a second row of two jobs that ran as long as the full one.)
                const uint32_t ngr0 = lane_of(ngr, 0), ng0 = lane_of(ng, 0);
                uint32_t Pc = min((lane & 15u) + 1u, pmax), perc = ngr0 ? ceil_div(ngr0, Pc) : 0u;
                if (Pc > 1u) perc |= 1u;     // (an odd number of groups per part keeps the parts out of each other's LDS banks: below)
                Pc = perc ? ceil_div(ngr0, perc) : 1u;
                const uint32_t rows_c = (ng0 * Pc + 63u) >> 6;
                uint32_t best = ((4u * rows_c * perc + (uint32_t)FFH_ROW_SETUP_Q * rows_c) << 16) | (rows_c << 8) | Pc;    // fewest row-steps (in quarters, + the rows' set-up), then fewest rows, then fewest parts
                best = min(best, (uint32_t)__builtin_amdgcn_update_dpp((int)best, (int)best, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
                best = min(best, (uint32_t)__builtin_amdgcn_update_dpp((int)best, (int)best, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
                best = min(best, (uint32_t)__builtin_amdgcn_update_dpp((int)best, (int)best, 0x141, 0xf, 0xf, false));   // row_half_mirror
                best = min(best, (uint32_t)__builtin_amdgcn_update_dpp((int)best, (int)best, 0x140, 0xf, 0xf, false));   // row_mirror
                P = lane_of(best, 0) & 0xFFu;
                per = ngr ? ceil_div(ngr, P) : 0u;
                if (P > 1u) per |= 1u;
            } else {
                P = (ngr + (uint32_t)kGroupsPerLane / 2u) / (uint32_t)kGroupsPerLane;   // ~ ngr / 6, rounded
                P = min(max(P, 1u), pmax);
                per = ngr ? ceil_div(ngr, P) : 0u;
                // The parts of one candidate are read by neighbouring lanes at the same time, `per` groups apart: with GW = 24 words per
                // group and an even `per` the parts p and p + 4 (per = 6: 144 p words) start in the same LDS bank with different
                // addresses -- a two-way conflict on every read of the row.  An odd number of groups per part spreads the parts.
                if (P > 1u) { per |= 1u; P = ceil_div(ngr, per); }
            }
            const uint32_t jobs = ng * P;
            uint32_t incl = jobs;   // inclusive scan over the row of 16 lanes
            incl += row_prev<1>(incl);
            incl += row_prev<2>(incl);
            incl += row_prev<4>(incl);
            incl += row_prev<8>(incl);
            // first slot of the bucket's part of the piece; a direct image: the database index of that slot
            const uint32_t dd = dd_off ? (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane + 16u) << 2), (int)dG) : 0u;
            // The tables hold the buckets that have jobs, in order; a job finds its bucket by counting the buckets that begin at or
            // before it: one marker bit per such bucket at the job BEFORE its first (the first bucket's jobs begin at 0 and need
            // none), so that "markers below my lane" + "markers of the rows before" is the table index (rows()).
            const bool ne = lane < 16u && jobs != 0u;
            const uint32_t slot = __builtin_amdgcn_mbcnt_lo((uint32_t)__builtin_amdgcn_ballot_w64(ne), 0u);
            const uint32_t js = incl - jobs;
            wave_lds_fence();   // (the markers are cleared)
            if (ne) {
                const uint32_t inv = (uint32_t)(65536.0f * __builtin_amdgcn_rcpf((float)P) + 0.999f);                 // ceil(65536 / P)
                W.tab[0][slot] = make_uint4(js, lo_c - c0, b0 + lane, P | (inv << 8));   // x / P == (x * inv) >> 16 for x < 4096, P <= 16
                W.tab[1][slot] = make_uint4((lo_g - g0) * GW, ngr, per, (lo_g << 5) + dd);
                if (js) atomicOr((unsigned long long *)&W.mark[(js - 1u) >> 6], 1ull << ((js - 1u) & 63u));
            }
            wave_lds_fence();
            return lane_of(incl, 15);
        };

        // every row of the parked piece's jobs
        auto rows = [&](uint32_t n_jobs) {
            uint32_t before = 0;   // buckets begun in the rows before this one
            for (uint32_t j0 = 0; j0 < n_jobs; j0 += 64) {
                const uint32_t J = j0 + lane;
                const uint64_t M = W.mark[j0 >> 6];
                const uint32_t mlo = (uint32_t)M, mhi = (uint32_t)(M >> 32);
                const uint32_t i = before + __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));   // the bucket of job J
                before += (uint32_t)__popc(uni(mlo)) + (uint32_t)__popc(uni(mhi));
                const uint4 ta = W.tab[0][i], tb = W.tab[1][i];
                const uint32_t jj = J - ta.x, P = ta.w & 31u;
                const uint32_t k = (jj * (ta.w >> 8)) >> 16, p = jj - k * P;                   // candidate k of the bucket, part p of it
                const bool valid = J < n_jobs;
                const uint32_t slot = min(ta.y + k, (uint32_t)kKC - 1u);
                const uint2 cand = W.cand[slot];
                const uint32_t gid = W.gid[slot];
                const uint32_t g_lo = p * tb.z;
                uint32_t trips = 0;
                if (valid && g_lo < tb.y) trips = min(tb.z, tb.y - g_lo);
                const uint32_t x = cand.y ^ ta.z;
                const uint32_t d = (uint32_t)__popc(((x >> rc.width) | x) & ((1u << rc.width) - 1u));   // mismatches inside the bucket key
                const uint32_t gword = tb.x + g_lo * GW, sbase = tb.w + (g_lo << 5);
                scan_row<RC, FARC, FFH_PIPE_TRIPS, side>(rc, cand.x, d, gid, trips, gword, sbase, W.strip);
            }
        };

        // ---- prologue: the work entries of four batches, candidate ids of two, groups / boundaries / guide entries of the first ----
        uint32_t q = next_q();
        if (q == kEnd) return;
        uint32_t E0 = load_entry(q), E1, E2, E3, E4 = 0, ends = 0;   // ends bit k: the entry k batches ahead is past the wave's last
        q = next_q(); E1 = load_entry(q); ends |= q == kEnd ? 2u : 0u;
        q = next_q(); E2 = load_entry(q); ends |= q == kEnd ? 4u : 0u;
        q = next_q(); E3 = load_entry(q); ends |= q == kEnd ? 8u : 0u;
        uint4 kreg[kKeyRegs];
        uint32_t greg_a[kGidRegs], greg_b[kGidRegs], dG0, dI0;
        uint2 ereg[kGidRegs];
#pragma unroll
        for (int j = 0; j < kKeyRegs; ++j) kreg[j] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < kGidRegs; ++j) { ereg[j] = make_uint2(0, 0); greg_a[j] = 0; greg_b[j] = 0; }
        const uint32_t cap_g = (uint32_t)kKW / GW;   // groups the strip holds
        {
            const uint32_t c0 = lane_of(E0, 4), c1 = lane_of(E0, 5), d0 = lane_of(E1, 4), d1 = lane_of(E1, 5), g0 = lane_of(E0, 2), g1 = lane_of(E0, 3);
            load_gids(c0, min(c1, c0 + (uint32_t)kKC), greg_a);
            load_gids(d0, min(d1, d0 + (uint32_t)kKC), greg_b);
            load_groups(g0, min(g1, g0 + cap_g), kreg);
            load_desc(E0, dG0, dI0);
            load_entries(c0, min(c1, c0 + (uint32_t)kKC), greg_a, ereg);
        }
        for (;;) {
            // The batch is worked off in PIECES of at most cap_g groups x kKC candidates: one piece, unless the entry does not fit the
            // strip (more candidates than it holds: a skewed guide set; correct for any input, the common case never takes a second
            // piece).  The first piece is what was requested a batch ago; further ones are fetched on the spot through the same
            // registers.  One copy of park() and rows() serves both (two inlined copies cost registers the kernel does not have).
            uint32_t t0 = lane_of(E0, 2), k0 = lane_of(E0, 4);   // the piece: groups from t0, candidates from k0
            for (bool first = true;; first = false) {
                const uint32_t b0 = lane_of(E0, 0), nbv = lane_of(E0, 1), g1 = lane_of(E0, 3), c0 = lane_of(E0, 4), c1 = lane_of(E0, 5);
                const uint32_t t1 = min(g1, t0 + cap_g), k1 = min(c1, k0 + (uint32_t)kKC);
                if (!first) {
                    if (k0 == c0) load_groups(t0, t1, kreg);
                    load_gids(k0, k1, greg_a);
                    load_entries(k0, k1, greg_a, ereg);
                }
                // everything requested a batch ago has arrived (the compiler's waits cover it): park it
                const uint32_t n_jobs = k1 > k0 ? park(b0, nbv, t0, t1, k0, k1, kreg, greg_a, ereg, dG0, dI0) : 0u;
                if (FFH_TRIP_STATS) ++st_parks;
                uint32_t nk = k0 + (uint32_t)kKC, nt = t0;
                if (nk >= c1) { nk = c0; nt = t0 + cap_g; }
                const bool last = nt >= g1;
                if (last) {
                    // ---- request what the next batches need: the work entry of batch +4, candidate ids of +2, boundaries, guide entries
                    //      and groups of +1 ----
                    q = next_q();
                    E4 = load_entry(q);
                    ends |= q == kEnd ? 16u : 0u;
                    const uint32_t d0 = lane_of(E2, 4), d1 = lane_of(E2, 5), n0 = lane_of(E1, 4), n1 = lane_of(E1, 5), h0 = lane_of(E1, 2), h1 = lane_of(E1, 3);
#pragma unroll
                    for (int j = 0; j < kGidRegs; ++j) greg_a[j] = greg_b[j];
                    load_gids(d0, min(d1, d0 + (uint32_t)kKC), greg_b);
                    load_entries(n0, min(n1, n0 + (uint32_t)kKC), greg_a, ereg);
                    load_groups(h0, min(h1, h0 + cap_g), kreg);
                    load_desc(E1, dG0, dI0);
                }
                // ---- compute this piece out of LDS ----
                if (n_jobs) rows(n_jobs);
                if (last) break;
                t0 = nt; k0 = nk;
            }
            if (ends & 2u) break;
            E0 = E1; E1 = E2; E2 = E3; E3 = E4;
            ends >>= 1;
        }
        stats_out();
    };
    run_side(std::integral_constant<int, 1>{});
    run_side(std::integral_constant<int, 0>{});
    if (FFH_TRIP_STATS && lane == 0) { atomicAdd(cursor + 28, (unsigned long long)hs.st_flush); atomicAdd(cursor + 29, (unsigned long long)hs.st_flush_it); }
    hs.finish();
}

// The instances.  Tuned ones -- far condition compiled in, queue and fixed-stride forms -- for the plans the cost model picks at genome
// scale for a 20-base pack (choose_plan / select_images: 11 + 9 with radii 2 + 1 for <= 4 mismatches, 10 + 10 with 1 + 1 and 2 + 2 for
// <= 3 and <= 5); for everything else one instance per PAIR of rest widths that can occur (prefix + suffix width = 20 or 19 compared
// bases, both rest keys 7 .. 12 bases), far condition read at run time, fixed stride.  (Round 3 had ONE any-width instance that chose
// the row form per row with a switch: its registers were the widest form's plus the switch's, 107 scalar and 8 vector registers
// spilled -- every small database, every 19-mer and Cpf1 scan ran on it.)
template <int R0, int R1, int FAR1>
inline void launch_compare_as(const CompareArgs &ca, unsigned long long *cursor, unsigned grid, hipStream_t st, int chunk) {
    if (chunk >= (int)kQueueChunkLong) hipLaunchKernelGGL((k_compare<R0, R1, FAR1, (int)kQueueChunkLong>), dim3(grid), dim3(kCmpThreads), 0, st, ca, cursor);
    else if (chunk > 0) hipLaunchKernelGGL((k_compare<R0, R1, FAR1, (int)kQueueChunkMedium>), dim3(grid), dim3(kCmpThreads), 0, st, ca, cursor);
    else hipLaunchKernelGGL((k_compare<R0, R1, FAR1, 0>), dim3(grid), dim3(kCmpThreads), 0, st, ca, cursor);
}
template <int R0, int R1>
inline void launch_compare_pair(const CompareArgs &ca, unsigned long long *cursor, unsigned grid, hipStream_t st) {
    hipLaunchKernelGGL((k_compare<R0, R1, -1, 0>), dim3(grid), dim3(kCmpThreads), 0, st, ca, cursor);
}
// Every pair of rest widths build_image_into accepts has an instance: prefix + suffix width = 20 compared bases -> (8,12) (9,11) (10,10)
// (11,9) (12,8); 19 compared bases -> (7,12) (8,11) (9,10) (10,9) (11,8) (12,7); a one-image plan takes the instance whose prefix form is
// its own.  The per-pair instances deal their work entries with a fixed stride whatever `chunk` / FFH_WORK_QUEUE say.
// chunk: the queue chunk both images' expected list lengths allow (work_list_chunk: two chunks per wave at least), 0 = fixed stride.
// Returns false for a pair of rest widths no instance exists for (the host refuses such images when they are built).
// generic_only / queue_env: the context's FFH_GENERIC_COMPARE / FFH_WORK_QUEUE switches (ffh_debug.hpp; tests: the per-pair instances for
// every plan; 0: never a queue, 1 / 16: chunks of 16, 4: chunks of 4, -1: by list length)
inline bool launch_compare(const CompareArgs &ca, unsigned long long *cursor, unsigned grid, hipStream_t st, int chunk, bool generic_only, int queue_env) {
    const int queue = queue_env < 0 ? chunk : queue_env == 1 ? (int)kQueueChunkLong : queue_env;
    const bool two = ca.side[1].n_list != nullptr;
    const int r0 = (int)ca.side[0].rest, far = two ? ca.side[1].r_far + 1 : 0;
    int r1 = two ? (int)ca.side[1].rest : 0;
    if (!generic_only && two) {
        if (r0 == 9 && r1 == 11 && far == 3) return launch_compare_as<9, 11, 3>(ca, cursor, grid, st, queue), true;
        if (r0 == 10 && r1 == 10 && far == 2) return launch_compare_as<10, 10, 2>(ca, cursor, grid, st, queue), true;
        if (r0 == 10 && r1 == 10 && far == 3) return launch_compare_as<10, 10, 3>(ca, cursor, grid, st, queue), true;
    }
    if (!two) r1 = r0 <= 8 ? 12 : 20 - r0;   // a one-image plan never runs the suffix side: any instance with this prefix form
    switch (r0 * 16 + r1) {
        case 7 * 16 + 12: return launch_compare_pair<7, 12>(ca, cursor, grid, st), true;
        case 8 * 16 + 12: return launch_compare_pair<8, 12>(ca, cursor, grid, st), true;
        case 8 * 16 + 11: return launch_compare_pair<8, 11>(ca, cursor, grid, st), true;
        case 9 * 16 + 11: return launch_compare_pair<9, 11>(ca, cursor, grid, st), true;
        case 9 * 16 + 10: return launch_compare_pair<9, 10>(ca, cursor, grid, st), true;
        case 10 * 16 + 10: return launch_compare_pair<10, 10>(ca, cursor, grid, st), true;
        case 10 * 16 + 9: return launch_compare_pair<10, 9>(ca, cursor, grid, st), true;
        case 11 * 16 + 9: return launch_compare_pair<11, 9>(ca, cursor, grid, st), true;
        case 11 * 16 + 8: return launch_compare_pair<11, 8>(ca, cursor, grid, st), true;
        case 12 * 16 + 8: return launch_compare_pair<12, 8>(ca, cursor, grid, st), true;
        case 12 * 16 + 7: return launch_compare_pair<12, 7>(ca, cursor, grid, st), true;
        default: return false;
    }
}
inline int work_list_chunk(double expected_entries, unsigned grid) {
    const double waves = (double)grid * kCmpWaves;
    return expected_entries >= 2.0 * kQueueChunkLong * waves ? (int)kQueueChunkLong : expected_entries >= 2.0 * kQueueChunkMedium * waves ? (int)kQueueChunkMedium : 0;
}

}  // namespace ffh
