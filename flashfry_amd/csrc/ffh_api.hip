// ffh_api.hip -- context, memory and launch orchestration behind the C ABI of include/flashfry_hip.h.
// gfx950 only.  One context = one GPU + one HIP stream; everything below is issued on that stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/flashfry_hip.h"
#include "ffh_debug.hpp"
#include "ffh_dbfile.hpp"
#include "ffh_ingest.hpp"
#include "ffh_inflate.hpp"
#include "ffh_kernels.hpp"
#include "ffh_compare.hpp"
#include "cfd_table.inc"
#include "jost_table.inc"

using namespace ffh;

static thread_local std::string g_create_error;  // what ffh_last_error(NULL) returns: per thread, contexts are created from several threads
namespace ffh {
void set_global_error(const std::string &m) { g_create_error = m; }
}  // namespace ffh

#define FFH_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                               \
            return FFH_E_HIP;                                                                           \
        }                                                                                               \
    } while (0)

namespace {

// A captured launch sequence (PrepGraph) holds raw pointers: it is only replayed while every buffer it refers to is where it was
// (prep_signature: address and capacity of each, per context).  While a sequence is being captured on this thread an allocation is
// refused (hipMalloc is not capturable): the caller then runs the sequence uncaptured.
static thread_local bool t_capturing = false;

template <typename T>
struct DevBuf {  // device allocation that grows on demand and frees itself (on the device that is current: the entry points set it)
    T *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    hipError_t reserve(size_t n) {  // contents are NOT preserved
        if (n <= cap) return hipSuccess;
        if (t_capturing) return hipErrorStreamCaptureUnsupported;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Image {  // one bucketed scan image of the database
    int width = -1;
    DevBuf<uint32_t> bstart;  // 4^width + 1: first target of every bucket
    DevBuf<uint32_t> gstart;  // 4^width + 1: first group of every bucket
    DevBuf<uint32_t> gwords;  // the bucket's targets in bit-sliced groups of 32 (ffh_compare.hpp), padded by kKW + 64 words
    DevBuf<uint32_t> tidx;    // database index of every slot (32 per group); not kept by a direct image
    bool direct = false;      // direct image (k_bucket_first): database index of a slot = slot + ddelta[bucket], ddelta = gstart + nb + 1
    uint32_t *ddelta() const { return gstart.p + ((size_t)1 << (2 * width)) + 1; }
    int rest = 0;             // bases in the rest key (the ones the bucket id does not hold)
    DevBuf<uint32_t> live;    // [2^live_bits] which bucket-id prefixes of live_bits = min(2 width, 12) bits hold a target (k_bucket_live)
    uint32_t live_bits = 0;
};

struct Plan { int a, r1, s, r2; };  // prefix width/radius, suffix width/radius (r2 < 0: no suffix pass)

struct Evt {
    hipEvent_t e = nullptr;
};

}  // namespace

// page-locked host blocks for results, recycled across calls (hipHostMalloc of several MB costs ~1 ms; pageable
// destinations make every device-to-host copy go through a bounce buffer)
// FFH_POOL_DEBUG=1 (tools/stress_parity.py): every block the pool hands out carries a canary over its slack [used, cap), checked when the
// block comes back (a write past a result's end); a block that comes back is filled with a poison pattern, checked when it is
// handed out again and when the pool dies (a device or host write into a block nobody owns: a late DMA, a stale pointer).
// ffh_debug_pool_errors() counts what the checks found.
static std::atomic<unsigned long long> g_pool_errors{0};
static bool all_bytes(const void *p, size_t n, unsigned char v) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) if (b[i] != v) return false;
    return true;
}
struct PinnedPool {
    static constexpr unsigned char kCanary = 0xA5, kPoison = 0xDB;
    std::mutex m;
    std::vector<std::pair<void *, size_t>> free_blocks;
    bool debug = false;       // FFH_POOL_DEBUG (ffh_debug.hpp), set when the context is created
    long limit_mb = 0;        // FFH_PINNED_LIMIT_MB: the most page-locked host memory ONE result block may take (page-locked memory is a
                              // resource the host shares with everything else on the node); a result that needs more fails with
                              // FFH_E_NOMEM instead of pinning it
    bool pool_debug() const { return debug; }
    void *get(size_t bytes, size_t &cap) {
        if (limit_mb > 0 && bytes > (size_t)limit_mb << 20) return nullptr;
        void *p = nullptr;
        {
            std::lock_guard<std::mutex> g(m);
            for (size_t i = 0; i < free_blocks.size(); ++i)
                if (free_blocks[i].second >= bytes && free_blocks[i].second <= 4 * bytes + (1u << 20)) {
                    p = free_blocks[i].first;
                    cap = free_blocks[i].second;
                    free_blocks.erase(free_blocks.begin() + (long)i);
                    break;
                }
        }
        if (p && pool_debug() && !all_bytes(p, cap, kPoison)) {
            g_pool_errors.fetch_add(1);
            fprintf(stderr, "[ffh pool debug] a released page-locked block (%zu bytes) was written to before it was handed out again\n", cap);
        }
        if (!p) {
            cap = bytes + bytes / 4 + 4096;
            if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
        }
        if (pool_debug()) std::memset((char *)p + bytes, kCanary, cap - bytes);
        return p;
    }
    void put(void *p, size_t cap, size_t used) {
        if (pool_debug()) {
            if (used <= cap && !all_bytes((const char *)p + used, cap - used, kCanary)) {
                g_pool_errors.fetch_add(1);
                fprintf(stderr, "[ffh pool debug] the slack behind a result block (%zu of %zu bytes used) was written to\n", used, cap);
            }
            std::memset(p, kPoison, cap);
        }
        std::lock_guard<std::mutex> g(m);
        if (free_blocks.size() >= 6) { check_poison(free_blocks.front()); (void)hipHostFree(free_blocks.front().first); free_blocks.erase(free_blocks.begin()); }
        free_blocks.emplace_back(p, cap);
    }
    void check_poison(const std::pair<void *, size_t> &b) const {
        if (pool_debug() && !all_bytes(b.first, b.second, kPoison)) {
            g_pool_errors.fetch_add(1);
            fprintf(stderr, "[ffh pool debug] a released page-locked block (%zu bytes) was written to before it was freed\n", b.second);
        }
    }
    ~PinnedPool() { for (auto &b : free_blocks) { check_poison(b); (void)hipHostFree(b.first); } }
};

struct ffh_result {
    uint32_t n_guides = 0;
    uint64_t n_hits = 0, n_positions = 0;
    bool offsets_pending = false;      // aggregates-only result: guide_offsets / n_hits are folded from the summaries when first asked for
    bool pos_offsets_pending = false;  // pos_offsets are folded from the counts in the hit target longs when first asked for
    std::once_flag offsets_once, pos_offsets_once;   // the accessors may be called from several host threads at once (the CLI formats rows in parallel)
    int scores_valid = 0;
    // the arrays live in two pinned blocks owned by the context's pool: everything per guide and per hit, and the positions
    // (whose number is known only after the per-hit arrays are on their way to the host)
    std::shared_ptr<PinnedPool> pool;
    void *block = nullptr, *block2 = nullptr;
    size_t block_cap = 0, block2_cap = 0, block_used = 0, block2_used = 0;
    ffh_guide_summary *summaries = nullptr;
    uint64_t *guide_offsets = nullptr, *hit_targets = nullptr, *pos_offsets = nullptr, *positions = nullptr;
    double *hit_cfd = nullptr;
    uint8_t *hit_mm = nullptr;

    // lists == false keeps only summaries + guide offsets
    bool allocate(const std::shared_ptr<PinnedPool> &p, uint32_t G, uint64_t H, bool lists, bool with_cfd = true, bool with_pos_offsets = true) {
        pool = p;
        n_guides = G; n_hits = H;
        auto up = [](size_t x) { return (x + 63) & ~(size_t)63; };
        size_t o_sum = 0, o_goff = o_sum + up((size_t)G * sizeof(ffh_guide_summary)), o_ht = o_goff + up(((size_t)G + 1) * 8);
        size_t o_cfd = o_ht, o_poff = o_ht, o_mm = o_ht, total = o_ht;
        if (lists) {
            o_cfd = o_ht + up((size_t)H * 8); o_poff = o_cfd + (with_cfd ? up((size_t)H * 8) : 0);
            o_mm = o_poff + (with_pos_offsets ? up(((size_t)H + 1) * 8) : 0); total = o_mm + up((size_t)H);
        }
        block_used = total + 64;
        block = pool->get(block_used, block_cap);
        if (!block) return false;
        char *b = (char *)block;
        summaries = (ffh_guide_summary *)(b + o_sum); guide_offsets = (uint64_t *)(b + o_goff);
        if (lists) {
            hit_targets = (uint64_t *)(b + o_ht); hit_mm = (uint8_t *)(b + o_mm);
            if (with_cfd) hit_cfd = (double *)(b + o_cfd);
            if (with_pos_offsets) pos_offsets = (uint64_t *)(b + o_poff);
        }
        return true;
    }
    bool allocate_positions(uint64_t P) {
        n_positions = P;
        block2_used = (size_t)P * 8 + 64;
        block2 = pool->get(block2_used, block2_cap);
        positions = (uint64_t *)block2;
        return block2 != nullptr;
    }
    ~ffh_result() {
        if (block && pool) pool->put(block, block_cap, block_used);
        if (block2 && pool) pool->put(block2, block2_cap, block2_used);
    }
};

struct ffh_ctx {
    int device = 0, enzyme = 0;
    hipStream_t st = nullptr;
    hipStream_t own_st = nullptr;  // the stream the context created; st may name the caller's instead (ffh_use_stream)
    hipStream_t copy_st = nullptr; // result copies to the host that run beside the kernels still producing the rest of the result
    hipEvent_t copy_ev = nullptr;
    bool borrowed = false;
    Geometry geo{};
    std::string err;

    // database
    uint64_t T = 0, P = 0;
    bool db_sorted = false;   // targets are in sequence order (every database the reference writes is)
    DevBuf<uint64_t> targets, positions, pos_off;
    Image img[2];  // 0 prefix, 1 suffix
    Image alt[2];  // a second pair of images with another split (select_images: 11 + 9 suits 4 mismatches at hg38 scale, 10 + 10 suits 5)
    bool auto_width = true;
    std::vector<std::string> contigs;
    std::vector<uint64_t> bin_bytes;
    uint32_t n_bins = 0, bin_begin = 0, bin_end = 0;
    double db_prepare_ms = 0;
    double span = 1.0;   // fraction of prefix-key space the shard's targets lie in (plan_cost)
    ffh_load_stats load{};
    double load_device_inflate_ms = 0;
    int plan_a = -1, plan_r1 = -1;
    // persistent waves: four 256-thread blocks per CU (LDS-limited), every wave walks its share of the batches
    unsigned compare_grid = 256 * 4;
    bool scan_timing_pending = false, finalize_timing_pending = false, hit_t_ready = false;
    uint32_t max_guide_batch = 0;  // 0 = as many guides per compare launch as the candidate list allows
    Switches sw;                   // the environment switches, read once by ffh_create (ffh_debug.hpp)
    bool too_many_hits = false;    // the last scan stopped at sw.raw_hit_limit raw hits: the caller splits the guide set (discover_split)

    // scan state
    DevBuf<uint64_t> guides;
    uint32_t n_guides = 0;
    int max_mm = 0;
    bool scanned = false;
    DevBuf<uint64_t> hits, hits_alt, hit_t;   // hit_t: target long of every raw hit, sorted order
    uint64_t *hits_sorted = nullptr;
    uint64_t n_raw = 0;
    int tbits = 1;   // hit key = (guide << tbits) | database index
    DevBuf<uint32_t> seg_begin, seg_end;
    // the two waits of a discover step -- for the compare launch's counters, for the epilogue's summaries -- poll a word in
    // page-locked memory that a one-wave kernel writes behind the work (k_publish): a hipStreamSynchronize wake-up costs 20-50 us
    // on this stack, which is 2-4 % of a 2.3 ms step.  Bounded spin, then the blocking call (spin_wait).
    unsigned long long *h_pub = nullptr, *d_pub = nullptr;   // [0..15] published counters, [16] sequence number
    unsigned long long pub_seq = 0;
    unsigned long long *d_counters = nullptr;  // [0] hit cursor, [1] pairs prefix, [2] pairs suffix, [3] a zero word, [4] load-time check counter

    // per-pass scratch
    DevBuf<uint2> gtab[2];                                  // {rest key, bucket} of every guide of the current batch, per side (L2-resident)
    DevBuf<uint32_t> gbucket[2], patterns[2], istart[2];
    DevBuf<unsigned long long> part_pairs[2];  // per candidate partition: targets x candidates of its buckets (k_item_bin)
    uint32_t n_part[2] = {0, 0};
    std::pair<int, int> patterns_key[2] = {{-1, -1}, {-1, -1}};  // (width, radius) of the pattern list resident in patterns[side]
    uint64_t db_gen = 0;        // moves on with every database load
    uint64_t pattern_gen = 0;   // moves on with every pattern upload: a captured launch sequence reads patterns[side] and must not outlive its content
    DevBuf<uint32_t> icount, ifill, item_gid, scan_tmp32;
    // candidate binning and work list of one image
    struct SideScratch { DevBuf<uint32_t> part_fill, part_hist, part_start, gp_start, by_part, scan_tmp; } side_scr[2];
    DevBuf<uint32_t> tmp_keys, tmp_tidx;                    // build_image's temporaries
    DevBuf<uint32_t> wl_count[2];                             // work entries per batch of buckets + per block of 1024 batches
    DevBuf<WorkEntry> wl_list[2];                             // the compare kernel's work list, per image
    DevBuf<uint64_t> scan_tmp64;
    DevBuf<uint32_t> sort_table, sort_offs, heavy_list;
    std::map<std::pair<int, int>, std::vector<uint32_t>> pattern_cache;

    // finalize scratch
    DevBuf<uint32_t> n_ret, ot_count, full, prior, out_cnt, out_tidx, totals, hit_pre;
    // bounded scan (ffh_scan_bounded): the suffix images of the slabs, the slabs' first targets, their prefix-bucket ranges, the
    // guides' running totals and the packed set of guides still active
    std::vector<std::unique_ptr<Image>> slab_img;
    std::vector<uint64_t> slab_t;
    DevBuf<uint32_t> g_total, g_flag, g_pos, g_map;
    DevBuf<uint64_t> g_active;
    int slabs_state = 0;      // 0 not built, 1 built, -1 this database cannot be bounded
    int bound_mode = 0;       // bounding on for this context
    bool bound_auto = true;   // ... switched on by the first scan that collects more than kBoundAutoHits raw hits per guide
    uint32_t bound_ot = 0;    // the limit the last scan was bounded by (0: it was not)
    DevBuf<uint64_t> ret_off, pos_base, out_target, out_posoff, out_pos;
    DevBuf<uint8_t> out_mm;
    DevBuf<double> out_cfd, out_hsu, out_jost;
    DevBuf<GuideSummary> summ, summ_stage;   // (summ_stage / ret_off_stage: what the copy stream reads of a pipelined call's first part)
    DevBuf<uint64_t> ret_off_stage;
    ScoreTables *d_tab = nullptr;

    // The candidate-list / work-list kernels of a scan (~26 launches of a few microseconds each: the host cannot issue them as fast
    // as the device runs them) as ONE captured graph, replayed while the call is the same in everything the launches depend on -- guide
    // buffer and count, plan, images, buffers (prep_signature), database and pattern generation.  The first call of a kind runs uncaptured (it may allocate), the second
    // captures, the following ones replay.  Work, results and counters are those of the plain launches; FFH_GRAPH=0 switches it off.
    struct PrepGraph {
        hipGraphExec_t exec = nullptr;
        uint64_t key[13] = {}, seen[13] = {}, epoch = 0, seen_epoch = 0;
        SideArgs side[2];
        double expect[2] = {0, 0};
        uint32_t n_part[2] = {0, 0};
    } pg_slots[2];        // [1]: the second part of a pipelined ffh_discover (another guide pointer and count: a sequence of its own)
    int pg_slot = 0;
    hipEvent_t ev[8] = {};
    ffh_timings tm{};
    std::shared_ptr<PinnedPool> pool = std::make_shared<PinnedPool>();
};

static uint32_t S_nb_plus_1(int width) { return (1u << (2 * width)) + 1u; }
static unsigned blocks_for(uint64_t n, unsigned threads) { return (unsigned)std::max<uint64_t>(1, (n + threads - 1) / threads); }

// ---- ball sizes and patterns -----------------------------------------------------------------------------
static double ball_size(int n, int r) {  // sum_k C(n,k) 3^k
    if (r < 0) return 0;
    double tot = 0, c = 1, p3 = 1;
    for (int k = 0; k <= std::min(n, r); ++k) {
        tot += c * p3;
        c = c * (n - k) / (k + 1);
        p3 *= 3;
    }
    return tot;
}

static void enum_patterns(int n, int r, int start, uint32_t cur, std::vector<uint32_t> &out) {
    out.push_back(cur);
    if (r == 0) return;
    for (int p = start; p < n; ++p)
        for (uint32_t d = 1; d <= 3; ++d)  // (dh, dl) in {01, 10, 11}: the three other bases
            enum_patterns(n, r - 1, p + 1, cur ^ (((d >> 1) << (n + p)) | ((d & 1u) << p)), out);
}

static const std::vector<uint32_t> &patterns_for(ffh_ctx *ctx, int n, int r) {
    r = std::min(r, n);
    auto key = std::make_pair(n, r);
    auto it = ctx->pattern_cache.find(key);
    if (it != ctx->pattern_cache.end()) return it->second;
    std::vector<uint32_t> v;
    v.reserve((size_t)ball_size(n, r));
    enum_patterns(n, r, 0, 0, v);
    std::sort(v.begin(), v.end());   // numeric order: patterns with equal high (partition) bits are neighbours (k_item_bin_direct)
    return ctx->pattern_cache.emplace(key, std::move(v)).first->second;
}

// cost of the best (r1, r2) for a prefix width a, in pair tests per guide: every candidate entry meets the slots of its bucket (the
// targets rounded up to groups of 32: + 16 on average) and costs about as much as kEntryCost pair tests to generate and bin
// (measured at hg38 scale: 10.4 ps per entry against 0.24 ps per pair test).
// span: the fraction of prefix-key space the shard's targets lie in (a bin shard of a database in sequence order is one
// contiguous range of it: 1/8 for one of eight shards).  Inside that range the prefix buckets are as full as the whole database's,
// outside it they are empty -- a candidate pattern that lands there is generated and looked up (about a third of an entry's cost) but
// neither binned nor compared; the suffix buckets thin out evenly.  Planning a shard by its target count alone took the 10 + 10
// split for an eighth of hg38 where 11 + 9, the whole database's plan, runs 37 % fewer pair tests.
static double plan_cost(double T, double span, int lc, int a, int max_mm, Plan &best) {
    constexpr double kEntryCost = 40.0;
    const int s = lc - a;
    const double per_p = span * (std::max(T / (span * std::pow(4.0, a)), 1.0) + 16.0) + kEntryCost * (1.0 + 2.0 * span) / 3.0;
    const double per_s = std::max(T / std::pow(4.0, s), 1.0) + 16.0 + kEntryCost;
    best = Plan{a, std::min(max_mm, a), s, -1};
    double best_cost = ball_size(a, best.r1) * per_p;
    if (max_mm >= 1)
        for (int r1 = 0; r1 <= std::min(max_mm - 1, a); ++r1) {
            const int r2 = max_mm - 1 - r1;
            if (r2 > s) continue;
            const double cost = ball_size(a, r1) * per_p + ball_size(s, r2) * per_s;
            if (cost < best_cost) { best_cost = cost; best = Plan{a, r1, s, r2}; }
        }
    return best_cost;
}
// the default prefix width: ~48 targets per (non-empty) prefix bucket, both keys <= 12 bases; a small database keeps the split near
// the middle (an image with more than 4^10 buckets costs more in bucket-proportional passes than its short candidate lists save)
static int default_prefix_width(double T, double span, int lc) {
    const int a = (int)std::floor(std::log(std::max(T, 1.0) / span / 48.0) / std::log(4.0));
    return std::max(lc - 12, std::min(12, std::max(a, lc - 10)));
}

static Plan choose_plan(const ffh_ctx *ctx, int max_mm) {
    const int lc = ctx->geo.lc, a = ctx->img[0].width, s = ctx->img[1].width;
    if (ctx->plan_r1 >= 0) {  // forced
        Plan p{a, std::min(ctx->plan_r1, a), s, max_mm - 1 - ctx->plan_r1};
        if (p.r1 >= max_mm || p.r1 >= a) { p.r1 = std::min(max_mm, a); p.r2 = -1; }
        if (p.r2 > s) p.r2 = s;
        return p;
    }
    Plan best;
    (void)plan_cost((double)std::max<uint64_t>(ctx->T, 1), ctx->span, lc, a, max_mm, best);
    if (a + s != lc) best = Plan{a, std::min(max_mm, a), s, -1};
    return best;
}

// ---- database residency ----------------------------------------------------------------------------------
// the image of targets [t_lo, t_lo + t_n) of the shard (the whole shard, or one slab of it in database order: ffh_scan_bounded)
static int build_image_into(ffh_ctx *ctx, Image &im, int which, int width, uint64_t t_lo, uint64_t t_n) {
    const uint32_t nb = 1u << (2 * width);
    im.width = width;
    im.rest = ctx->geo.lc - width;
    if (im.rest < kMinRest || im.rest > kMaxRest) {   // the compare kernel has one row form per rest width (ffh_compare.hpp)
        ctx->err = "bucket width " + std::to_string(width) + " leaves a rest key of " + std::to_string(im.rest) + " bases; the scan supports " +
                   std::to_string(kMinRest) + " .. " + std::to_string(kMaxRest);
        im.width = -1;
        return FFH_E_ARG;
    }
    const uint32_t R = (uint32_t)im.rest, GW = (uint32_t)group_words(im.rest);
    // every bucket rounds its targets up to whole groups of 32: at most T / 32 + nb groups (no host round trip for the exact number)
    const uint64_t max_groups = t_n / 32 + nb;
    if (max_groups * 32 >= (1ull << 31) - 64) { ctx->err = "too many target slots in one shard; split the bins across more GPUs"; return FFH_E_ARG; }
    DevBuf<uint32_t> &keys = ctx->tmp_keys, &tidx_in = ctx->tmp_tidx;   // the counting sort's output, bit-sliced below (shared by the two images:
                                                                         // allocating and freeing GB-sized buffers costs tens of ms each)
    // a prefix image over a 3'-PAM database in sequence order keeps its buckets in database order: no slot -> index array (k_bucket_first)
    im.direct = !ctx->sw.no_direct && which == 0 && ctx->geo.c0 != 0 && ctx->db_sorted;
    FFH_HIP(im.bstart.reserve((size_t)nb + 1));
    FFH_HIP(im.gstart.reserve(2 * ((size_t)nb + 1)));   // (+ ddelta behind it)
    FFH_HIP(im.gwords.reserve((size_t)max_groups * GW + kKW + 64));
    if (im.direct) im.tidx.release();
    else {
        FFH_HIP(keys.reserve(t_n + 1));
        FFH_HIP(tidx_in.reserve(t_n + 1));
        FFH_HIP(im.tidx.reserve((size_t)max_groups * 32 + 64));
    }
    FFH_HIP(ctx->icount.reserve((size_t)nb + 1));
    FFH_HIP(ctx->ifill.reserve((size_t)nb + 1));
    FFH_HIP(ctx->scan_tmp32.reserve(scan_scratch_elems_safe(nb)));
    FFH_HIP(hipMemsetAsync(ctx->icount.p, 0, ((size_t)nb + 1) * 4, ctx->st));
    FFH_HIP(hipMemsetAsync(ctx->ifill.p, 0, ((size_t)nb + 1) * 4, ctx->st));
    const unsigned bl = blocks_for(t_n, 256);
    const uint64_t *tg = ctx->targets.p + t_lo;
    if (t_n) {
        if (which == 0) hipLaunchKernelGGL(k_image_hist<false>, dim3(bl), dim3(256), 0, ctx->st, tg, t_n, ctx->geo, width, ctx->icount.p);
        else hipLaunchKernelGGL(k_image_hist<true>, dim3(bl), dim3(256), 0, ctx->st, tg, t_n, ctx->geo, width, ctx->icount.p);
    }
    exclusive_scan<uint32_t, uint32_t>(ctx->icount.p, nb, im.bstart.p, ctx->scan_tmp32.p, ctx->st);
    if (t_n && im.direct) hipLaunchKernelGGL(k_bucket_first, dim3(bl), dim3(256), 0, ctx->st, tg, t_n, ctx->geo, width, ctx->ifill.p);
    if (t_n && !im.direct) {
        if (which == 0) hipLaunchKernelGGL(k_image_scatter<false>, dim3(bl), dim3(256), 0, ctx->st, tg, t_n, ctx->geo, width, im.bstart.p, ctx->ifill.p, keys.p, tidx_in.p, (uint32_t)t_lo);
        else hipLaunchKernelGGL(k_image_scatter<true>, dim3(bl), dim3(256), 0, ctx->st, tg, t_n, ctx->geo, width, im.bstart.p, ctx->ifill.p, keys.p, tidx_in.p, (uint32_t)t_lo);
    }
    {   // the partitions that hold a target: candidate entries of the others are dropped before they are enumerated
        im.live_bits = (uint32_t)std::min(2 * width, kMaxPartBits);
        FFH_HIP(im.live.reserve((size_t)1 << im.live_bits));
        FFH_HIP(hipMemsetAsync(im.live.p, 0, ((size_t)4) << im.live_bits, ctx->st));
        hipLaunchKernelGGL(k_bucket_live, dim3(blocks_for(nb, 256)), dim3(256), 0, ctx->st, im.bstart.p, nb, (uint32_t)(2 * width) - im.live_bits, im.live.p);
    }
    hipLaunchKernelGGL(k_group_count, dim3(blocks_for(nb, 256)), dim3(256), 0, ctx->st, im.bstart.p, nb, ctx->icount.p);
    exclusive_scan<uint32_t, uint32_t>(ctx->icount.p, nb, im.gstart.p, ctx->scan_tmp32.p, ctx->st);
    if (im.direct)   // (ifill holds the buckets' first database indices; a slab never builds a prefix image, so t_lo is 0 here)
        hipLaunchKernelGGL(k_group_build_direct, dim3(blocks_for(nb, 4)), dim3(256), 0, ctx->st, im.bstart.p, im.gstart.p, (const uint32_t *)ctx->ifill.p, tg, ctx->geo, width,
                           nb, R, GW, im.gwords.p, im.ddelta());
    else hipLaunchKernelGGL(k_group_build, dim3(blocks_for(nb, 4)), dim3(256), 0, ctx->st, im.bstart.p, im.gstart.p, keys.p, tidx_in.p, nb, R, GW, im.gwords.p, im.tidx.p);
    FFH_HIP(hipGetLastError());
    return FFH_OK;
}

static void drop_slabs(ffh_ctx *ctx);
static int build_image(ffh_ctx *ctx, int which, int width) { return build_image_into(ctx, ctx->img[which], which, width, 0, ctx->T); }

// targets/positions are already on the device in ctx->targets / ctx->positions
static int prepare_database(ffh_ctx *ctx) {
    if (ctx->T >= (1ull << 31) - 64) { ctx->err = "more than 2^31 targets in one shard; split the bins across more GPUs"; return FFH_E_ARG; }
    ++ctx->db_gen;   // (a new database: nothing captured against the old one may be replayed)
    FFH_HIP(hipEventRecord(ctx->ev[0], ctx->st));
    // counts -> position offsets; validate counts like BlockManager.scala:232-236
    FFH_HIP(ctx->out_cnt.reserve(ctx->T + 1));
    FFH_HIP(ctx->pos_off.reserve(ctx->T + 1));
    FFH_HIP(ctx->scan_tmp64.reserve(scan_scratch_elems_safe(ctx->T)));
    uint32_t *bad = (uint32_t *)ctx->d_counters + 16;   // two words: bad counts, neighbours out of sequence order
    FFH_HIP(hipMemsetAsync(bad, 0, 8, ctx->st));
    if (ctx->T) hipLaunchKernelGGL(k_check_counts, dim3(blocks_for(ctx->T, 256)), dim3(256), 0, ctx->st, ctx->targets.p, ctx->T,
                                  (1ull << (2 * ctx->geo.scan_len)) - 1ull, ctx->out_cnt.p, bad);
    exclusive_scan<uint32_t, uint64_t>(ctx->out_cnt.p, ctx->T, ctx->pos_off.p, ctx->scan_tmp64.p, ctx->st);
    uint32_t hbad2[2] = {0, 0};
    uint64_t total = 0;
    FFH_HIP(hipMemcpyAsync(hbad2, bad, 8, hipMemcpyDeviceToHost, ctx->st));
    FFH_HIP(hipMemcpyAsync(&total, ctx->pos_off.p + ctx->T, 8, hipMemcpyDeviceToHost, ctx->st));
    FFH_HIP(hipStreamSynchronize(ctx->st));
    const uint32_t hbad = hbad2[0];
    ctx->db_sorted = hbad2[1] == 0;
    if (hbad) { ctx->err = "Encoded position count should be greater than zero (and fit a signed short)"; return FFH_E_FORMAT; }
    if (total != ctx->P) { ctx->err = "positions array length does not equal the sum of the target counts"; return FFH_E_FORMAT; }
    // the part of prefix-key space the shard covers (plan_cost): first and last target of a database in sequence order
    ctx->span = 1.0;
    const int lc = ctx->geo.lc;
    if (ctx->db_sorted && ctx->geo.c0 + lc == ctx->geo.scan_len && ctx->T >= 2 && lc >= 12) {   // (a 3' PAM: the compared bases lead the sequence)
        uint64_t ends[2];
        FFH_HIP(hipMemcpyAsync(&ends[0], ctx->targets.p, 8, hipMemcpyDeviceToHost, ctx->st));
        FFH_HIP(hipMemcpyAsync(&ends[1], ctx->targets.p + (ctx->T - 1), 8, hipMemcpyDeviceToHost, ctx->st));
        FFH_HIP(hipStreamSynchronize(ctx->st));
        const int sh = 2 * (ctx->geo.scan_len - 12);
        const uint64_t k0 = (ends[0] >> sh) & 0xFFFFFFull, k1 = (ends[1] >> sh) & 0xFFFFFFull;
        if (k1 >= k0) ctx->span = std::min(1.0, std::max((double)(k1 - k0 + 1) / 16777216.0, 1.0 / 4096.0));
    }
    int a = ctx->plan_a >= 0 ? std::max(lc - 12, std::min(12, ctx->plan_a)) : default_prefix_width((double)ctx->T, ctx->span, lc);
    drop_slabs(ctx);   // (slab images of the database that was resident before)
    ctx->alt[0] = Image(); ctx->alt[1] = Image();
    int rc = build_image(ctx, 0, a);
    if (rc) return rc;
    rc = build_image(ctx, 1, lc - a);
    if (rc) return rc;
    FFH_HIP(hipEventRecord(ctx->ev[1], ctx->st));
    FFH_HIP(hipStreamSynchronize(ctx->st));
    ctx->tmp_keys.release(); ctx->tmp_tidx.release();   // (after the timed region)
    float ms = 0;
    FFH_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->db_prepare_ms = ms;
    ctx->scanned = false;
    return FFH_OK;
}

// ---- candidate lists (CSR) + work items of one image (no host synchronisation: counts stay on the device) ------
// im: the image the candidates are for (the shard's, or one slab's: ffh_scan_bounded; entries of its partitions without a target
// are dropped); gptr: the ng guides of this launch; seg_at >= 0: also clear the hit segments of guides
// seg_at .. seg_at + ng - 1 (their numbers in the caller's array)
static int prepare_side(ffh_ctx *ctx, hipStream_t st, int which, const Image &im, int radius, const uint64_t *gptr, int64_t seg_at, uint32_t ng,
                        uint32_t item_base, uint32_t rank_lo = 0u, uint32_t rank_hi = 63u) {
    const int width = im.width;
    const uint32_t nb = 1u << (2 * width);
    const std::vector<uint32_t> &pat = patterns_for(ctx, width, radius);
    const uint32_t np = (uint32_t)pat.size();
    ffh_ctx::SideScratch &sc = ctx->side_scr[which];
    DevBuf<uint32_t> &patterns = ctx->patterns[which], &gbucket = ctx->gbucket[which], &istart = ctx->istart[which];
    if (ctx->patterns_key[which] != std::make_pair(width, std::min(radius, width)) || patterns.cap < np) {  // uploaded once per (width, radius)
        FFH_HIP(patterns.reserve(np));
        FFH_HIP(hipMemcpyAsync(patterns.p, pat.data(), (size_t)np * 4, hipMemcpyHostToDevice, st));
        ctx->patterns_key[which] = std::make_pair(width, std::min(radius, width));
        ++ctx->pattern_gen;
    }
    FFH_HIP(gbucket.reserve(ng));
    FFH_HIP(ctx->gtab[which].reserve((size_t)ng + 64));
    FFH_HIP(istart.reserve((size_t)nb + 1));
    FFH_HIP(sc.scan_tmp.reserve(scan_scratch_elems_safe(nb)));
    // exact binning of the implicit (bucket, guide) entries into CSR form (see ffh_kernels.hpp)
    ItemGeom ig;
    ig.n_guides = ng; ig.n_pat = np;
    // 1024 buckets per partition keep a partition's candidate ids (~13k at hg38 scale) inside the 56 KB LDS stage of k_item_bin,
    // which lets two of its blocks share a CU; 12-base images (24-bit bucket ids) need 4096 per partition to stay within 4096 partitions
    ig.low_bits = (uint32_t)std::max(std::min(2 * width, kMaxLowBits - 2), 2 * width - kMaxPartBits);
    // ... and more, smaller partitions when that stage would overflow on average (a 10-base image at 5 mismatches: 4.4e7 entries in
    // 1024 partitions took k_item_bin's two-pass path for every partition)
    while (ig.low_bits > 2u && 2u * (uint32_t)width - (ig.low_bits - 1u) <= (uint32_t)kMaxPartBits &&
           (double)ng * (double)np / (double)(1u << (2u * (uint32_t)width - ig.low_bits)) > 0.9 * (double)kBinStage)
        --ig.low_bits;
    const uint32_t part_bits = 2u * (uint32_t)width - ig.low_bits;
    if (part_bits > (uint32_t)kMaxPartBits || ig.low_bits > (uint32_t)kMaxLowBits) { ctx->err = "bucket width too large for the candidate binning"; return FFH_E_ARG; }
    ig.n_part = 1u << part_bits;
    ig.item_base = item_base;
    ig.live = im.live.p; ig.live_bits = im.live_bits;
    ig.rank_lo = rank_lo; ig.rank_hi = rank_hi; ig.width = (uint32_t)width;
    const bool filtered = rank_lo > 0u || rank_hi < 63u;   // one slab of a bounded scan: the sizes are counted, not derived
    FFH_HIP(sc.part_hist.reserve((size_t)ig.n_part + 1));
    FFH_HIP(ctx->part_pairs[which].reserve((size_t)ig.n_part + 1));
    // the launch also clears the partition histogram and, on the prefix side, the guides' hit segments (one thread per guide anyway:
    // saves the fill launches before k_guide_part_hist and k_segments)
    if (which == 0) hipLaunchKernelGGL(k_guide_keys<false>, dim3(blocks_for(ng, 256)), dim3(256), 0, st, gptr, ng, ctx->geo, width, ctx->gtab[0].p, gbucket.p,
                                       seg_at >= 0 ? ctx->seg_begin.p + seg_at : (uint32_t *)nullptr, seg_at >= 0 ? ctx->seg_end.p + seg_at : (uint32_t *)nullptr,
                                       sc.part_hist.p, ig.n_part);
    else hipLaunchKernelGGL(k_guide_keys<true>, dim3(blocks_for(ng, 256)), dim3(256), 0, st, gptr, ng, ctx->geo, width, ctx->gtab[1].p, gbucket.p, (uint32_t *)nullptr,
                            (uint32_t *)nullptr, sc.part_hist.p, ig.n_part);
    FFH_HIP(sc.part_fill.reserve((size_t)2 * ig.n_part + 2));
    FFH_HIP(sc.part_start.reserve((size_t)ig.n_part + 2));
    uint32_t *part_count = sc.part_fill.p, *part_fill = sc.part_fill.p + ig.n_part + 1;
    hipLaunchKernelGGL(k_guide_part_hist, dim3(kPartHistBlocks), dim3(1024), 0, st, gbucket.p, ng, ig.low_bits, ig.n_part, sc.part_hist.p, sc.part_fill.p, 2u * ig.n_part + 2u);
    // guides grouped by partition (counting sort on the histogram), then every partition's block enumerates its own entries from
    // those runs: no intermediate records (ffh_kernels.hpp: k_item_bin_direct)
    FFH_HIP(sc.gp_start.reserve((size_t)ig.n_part + 2));
    FFH_HIP(sc.by_part.reserve((size_t)ng + 1));
    hipLaunchKernelGGL(k_guide_by_part, dim3(blocks_for(ng, 1024)), dim3(1024), 0, st, (const uint32_t *)gbucket.p, ng, ig.low_bits, ig.n_part,
                       (const uint32_t *)sc.part_hist.p, sc.gp_start.p, part_fill, sc.by_part.p);
    if (!filtered) hipLaunchKernelGGL(k_part_sizes, dim3(blocks_for(ig.n_part, 4)), dim3(256), 0, st, sc.part_hist.p, patterns.p, ig, part_bits, part_count);
    else hipLaunchKernelGGL((k_item_bin_direct<true, true>), dim3(ig.n_part), dim3(kPartThreads), 0, st, (const uint32_t *)sc.gp_start.p, (const uint32_t *)sc.by_part.p,
                            (const uint32_t *)patterns.p, ig, part_bits, (const uint32_t *)nullptr, part_count, (uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                            (unsigned long long *)nullptr);
    exclusive_scan<uint32_t, uint32_t>(part_count, ig.n_part, sc.part_start.p, sc.scan_tmp.p, st);
    if (!filtered) hipLaunchKernelGGL((k_item_bin_direct<false, false>), dim3(ig.n_part), dim3(kPartThreads), 0, st, (const uint32_t *)sc.gp_start.p, (const uint32_t *)sc.by_part.p,
                                      (const uint32_t *)patterns.p, ig, part_bits, (const uint32_t *)sc.part_start.p, (uint32_t *)nullptr, istart.p, ctx->item_gid.p,
                                      (const uint32_t *)im.bstart.p, ctx->part_pairs[which].p);
    else hipLaunchKernelGGL((k_item_bin_direct<false, true>), dim3(ig.n_part), dim3(kPartThreads), 0, st, (const uint32_t *)sc.gp_start.p, (const uint32_t *)sc.by_part.p,
                            (const uint32_t *)patterns.p, ig, part_bits, (const uint32_t *)sc.part_start.p, (uint32_t *)nullptr, istart.p, ctx->item_gid.p, (const uint32_t *)im.bstart.p,
                            ctx->part_pairs[which].p);
    ctx->n_part[which] = ig.n_part;
    FFH_HIP(hipGetLastError());
    return FFH_OK;
}

// Device buffers of the caller are produced and consumed by the caller's streams (a tensor fill, an RCCL collective).  With the
// context on its own stream the entry points that touch them wait for the device before and for the stream after their kernels;
// on the caller's stream (ffh_use_stream) stream order does the same for free.
static hipError_t fence_in(ffh_ctx *ctx) { return ctx->borrowed ? hipSuccess : hipDeviceSynchronize(); }
static hipError_t fence_out(ffh_ctx *ctx) { return ctx->borrowed ? hipSuccess : hipStreamSynchronize(ctx->st); }

// copies the counter block to page-locked memory and, after it, the sequence number the host is polling for
__global__ void k_publish(const unsigned long long *__restrict__ counters, volatile unsigned long long *__restrict__ host, unsigned long long seq,
                          const uint32_t *__restrict__ word /* nullable: one more device word the host wants -> host[17] */) {
    if (counters && threadIdx.x < 16) host[threadIdx.x] = counters[threadIdx.x];
    if (word && threadIdx.x == 17) host[17] = *word;
    __threadfence_system();
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) host[16] = seq;
}
// everything issued on the stream so far has completed (and `out`, if given, holds the device counters; `word_out` the device word
// `word`).  Whatever the host wants to read after the wait has to come through the page-locked block: an asynchronous copy into
// pageable host memory -- a stack variable -- is only known to have landed after a stream synchronisation, not when a later kernel's
// store is seen (the number of guides still active after a slab was read that way; once in ~60 000 randomised cases it was stale).
static hipError_t spin_wait(ffh_ctx *ctx, unsigned long long *out /* 16 words, nullable */, const uint32_t *word = nullptr, uint32_t *word_out = nullptr) {
    if (ctx->sw.no_spin || !ctx->h_pub) {
        if (out) { hipError_t e = hipMemcpyAsync(out, ctx->d_counters, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->st); if (e != hipSuccess) return e; }
        if (word) { hipError_t e = hipMemcpyAsync(word_out, word, 4, hipMemcpyDeviceToHost, ctx->st); if (e != hipSuccess) return e; }
        return hipStreamSynchronize(ctx->st);
    }
    const unsigned long long seq = ++ctx->pub_seq;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ctx->st, out ? (const unsigned long long *)ctx->d_counters : (const unsigned long long *)nullptr,
                       (volatile unsigned long long *)ctx->d_pub, seq, word);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    volatile unsigned long long *h = ctx->h_pub;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 0; h[16] != seq; ++it) {
        __builtin_ia32_pause();
        if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {   // a long kernel, a fault: block
            e = hipStreamSynchronize(ctx->st);
            if (e != hipSuccess) return e;
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (out) for (int i = 0; i < 16; ++i) out[i] = h[i];
    if (word) *word_out = (uint32_t)h[17];
    return hipSuccess;
}

template <typename T>
static hipError_t grow_keep(DevBuf<T> &b, size_t used, size_t need, hipStream_t st) {  // like reserve, but the first `used` elements survive
    if (need <= b.cap) return hipSuccess;
    DevBuf<T> nb;
    hipError_t e = nb.reserve(std::max(need, b.cap + b.cap / 2));
    if (e != hipSuccess) return e;
    if (used) e = hipMemcpyAsync(nb.p, b.p, used * sizeof(T), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { nb.release(); return e; }
    b = std::move(nb);
    return hipSuccess;
}

// =============================================================================================================
// C ABI
// =============================================================================================================
extern "C" {

int ffh_version(void) { return FFH_VERSION; }
unsigned long long ffh_debug_pool_errors(void) { return g_pool_errors.load(); }

int ffh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *ffh_last_error(const ffh_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static bool set_enzyme(ffh_ctx *ctx, int enzyme_index) {
    static const struct { int c0, lc, scan, cas9_23; } G[7] = {
        {0, 0, 0, 0},
        {0, 20, 24, 0},  // 1 Cpf1: comparisonBitEncoding 0x00FFFFFFFFFF, StandardScanParameters.scala:205
        {3, 20, 23, 1},  // 2 spCas9 0x3FFFFFFFFFC0 :99
        {3, 20, 23, 1},  // 3 NGG :143
        {3, 20, 23, 1},  // 4 NAG :187
        {3, 19, 22, 0},  // 5 19-mer 0x0FFFFFFFFFC0 :121
        {3, 19, 22, 0},  // 6 NGG 19-mer :165
    };
    if (enzyme_index < 1 || enzyme_index > 6) return false;
    ctx->enzyme = enzyme_index;
    ctx->geo = Geometry{G[enzyme_index].c0, G[enzyme_index].lc, G[enzyme_index].scan, G[enzyme_index].cas9_23};
    return true;
}

ffh_ctx *ffh_create(int device_id, int enzyme_index) {
    if (enzyme_index < 0 || enzyme_index > 6) { g_create_error = "Unable to find the correct parameter pack for enzyme: " + std::to_string(enzyme_index); return nullptr; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_create_error = "no HIP device available (flashfry_hip has no CPU fallback)"; return nullptr; }
    if (device_id < 0 || device_id >= n) { g_create_error = "device id out of range"; return nullptr; }
    ffh_ctx *ctx = new (std::nothrow) ffh_ctx();
    if (!ctx) { g_create_error = "out of memory"; return nullptr; }
    ctx->device = device_id;
    if (enzyme_index) set_enzyme(ctx, enzyme_index);  // 0: taken from the database header by ffh_db_open / ffh_db_open_header
    ctx->sw = Switches::from_env();
    if (ctx->sw.compare_grid) ctx->compare_grid = ctx->sw.compare_grid;
    ctx->max_guide_batch = ctx->sw.max_guide_batch;
    ctx->pool->debug = ctx->sw.pool_debug; ctx->pool->limit_mb = ctx->sw.pinned_limit_mb;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_st, hipStreamNonBlocking);
    ctx->st = ctx->own_st;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->copy_ev, hipEventDisableTiming);
    for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreate(&ctx->ev[i]);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_counters, kCounterWords * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_pub, 32 * sizeof(unsigned long long), hipHostMallocMapped);
    if (e == hipSuccess) { std::memset(ctx->h_pub, 0, 32 * sizeof(unsigned long long)); e = hipHostGetDevicePointer((void **)&ctx->d_pub, ctx->h_pub, 0); }
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_tab, sizeof(ScoreTables));
    if (e == hipSuccess) {
        ScoreTables h;
        std::memcpy(h.cfd_mm, FFH_CFD_MM, sizeof h.cfd_mm);
        std::memcpy(h.cfd_pam, FFH_CFD_PAM, sizeof h.cfd_pam);
        static const double coeff[20] = {0.0, 0.0, 0.014, 0.0, 0.0, 0.395, 0.317, 0.0, 0.389, 0.079,   // CrisprMitEduOffTarget.scala:43-47
                                         0.445, 0.508, 0.613, 0.851, 0.732, 0.828, 0.615, 0.804, 0.685, 0.583};
        std::memcpy(h.hsu_coeff, coeff, sizeof coeff);
        std::memcpy(h.jost, FFH_JOST, sizeof h.jost);
        e = hipMemcpy(ctx->d_tab, &h, sizeof h, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        g_create_error = std::string("HIP initialisation failed: ") + hipGetErrorString(e);
        ffh_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

void ffh_destroy(ffh_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->st);
    for (auto &g : ctx->pg_slots) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_tab) (void)hipFree(ctx->d_tab);
    if (ctx->h_pub) (void)hipHostFree(ctx->h_pub);
    for (auto &e : ctx->ev) if (e) (void)hipEventDestroy(e);
    if (ctx->copy_ev) (void)hipEventDestroy(ctx->copy_ev);
    if (ctx->copy_st) { (void)hipStreamSynchronize(ctx->copy_st); (void)hipStreamDestroy(ctx->copy_st); }
    if (ctx->own_st) (void)hipStreamDestroy(ctx->own_st);
    delete ctx;  // the device buffers free themselves
}

int ffh_set_plan(ffh_ctx *ctx, int prefix_bases, int prefix_radius) {
    if (!ctx) return FFH_E_ARG;
    if (prefix_bases > 12) { ctx->err = "prefix_bases must be <= 12"; return FFH_E_ARG; }
    const bool rebuild = ctx->T && prefix_bases >= 0 && prefix_bases != ctx->img[0].width;
    ctx->plan_a = prefix_bases;
    ctx->plan_r1 = prefix_radius;
    if (rebuild) {
        (void)hipSetDevice(ctx->device);
        return prepare_database(ctx);
    }
    return FFH_OK;
}

int ffh_db_load_soa(ffh_ctx *ctx, const uint64_t *targets, uint64_t n_targets, const uint64_t *positions, uint64_t n_positions, int on_device) {
    if (!ctx || (n_targets && !targets) || (n_positions && !positions)) { if (ctx) ctx->err = "null argument"; return FFH_E_ARG; }
    if (ctx->enzyme == 0) { ctx->err = "the context has no enzyme yet: create it with an enzyme index or open a database file"; return FFH_E_STATE; }
    FFH_HIP(hipSetDevice(ctx->device));
    if (on_device) FFH_HIP(hipDeviceSynchronize());  // the producer (e.g. torch) used another stream
    ctx->T = n_targets;
    ctx->P = n_positions;
    FFH_HIP(ctx->targets.reserve(n_targets + 1));
    FFH_HIP(ctx->positions.reserve(n_positions + 1));
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (n_targets) FFH_HIP(hipMemcpyAsync(ctx->targets.p, targets, n_targets * 8, kind, ctx->st));
    if (n_positions) FFH_HIP(hipMemcpyAsync(ctx->positions.p, positions, n_positions * 8, kind, ctx->st));
    FFH_HIP(hipStreamSynchronize(ctx->st));
    return prepare_database(ctx);
}

// raw bin payloads already on the device -> ctx->targets / ctx->positions (ffh_ingest.hpp), then the scan images
static int decode_blocks_on_device(ffh_ctx *ctx, const int64_t *d_raw, const std::vector<uint64_t> &bin_off, const std::vector<uint64_t> &bin_len) {
    const uint32_t nb = (uint32_t)bin_off.size();
    hipStream_t st = ctx->st;
    DevBuf<uint64_t> d_off, d_pbase, d_scr64;
    DevBuf<uint32_t> d_hdr, d_plen, d_marks, d_rank, d_scr32;
    std::vector<uint64_t> ends(nb + 1, 0);  // the kernels take off[b] .. off[b + 1]: the bins must lie back to back, as DatabaseWriter.scala:75-92 writes them
    for (uint32_t b = 0; b < nb; ++b) {
        ends[b] = bin_off[b];
        if (b + 1 < nb && bin_off[b] + bin_len[b] != bin_off[b + 1]) { ctx->err = "bin payloads are not stored back to back"; return FFH_E_FORMAT; }
    }
    ends[nb] = nb ? bin_off[nb - 1] + bin_len[nb - 1] : 0;
    FFH_HIP(d_off.reserve(nb + 1));
    FFH_HIP(d_pbase.reserve(nb + 2));
    FFH_HIP(d_scr64.reserve(scan_scratch_elems_safe(nb + 1)));
    FFH_HIP(d_hdr.reserve(nb + 1));
    FFH_HIP(d_plen.reserve(nb + 8));
    unsigned long long *d_err = ctx->d_counters + 10;
    FFH_HIP(hipMemcpyAsync(d_off.p, ends.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, st));
    FFH_HIP(hipMemsetAsync(d_err, 0xFF, 8, st));
    FFH_HIP(hipMemsetAsync(d_plen.p, 0, (size_t)(nb + 8) * 4, st));
    if (nb) hipLaunchKernelGGL(k_block_heads, dim3(blocks_for(nb, 256)), dim3(256), 0, st, d_raw, d_off.p, nb, d_hdr.p, d_plen.p, d_err);
    exclusive_scan<uint32_t, uint64_t>(d_plen.p, nb, d_pbase.p, d_scr64.p, st);
    unsigned long long herr = ~0ull;
    uint64_t n_payload = 0;
    auto report = [&](unsigned long long key) -> int {
        const uint32_t bin = (uint32_t)(key >> 36), code = (uint32_t)(key & 15u);
        switch (code) {
            case kBlkEmpty: ctx->err = "empty block for bin " + std::to_string(bin); break;
            case kBlkShortTable: ctx->err = "indexed block shorter than its lookup table"; break;
            case kBlkNotContiguous: ctx->err = "indexed block: sub-bin table is not contiguous"; break;
            case kBlkSliceRange: ctx->err = "indexed block: sub-bin slice out of range"; break;
            case kBlkCover: ctx->err = "indexed block: sub-bin sizes do not cover the payload"; break;
            case kBlkType: {
                int64_t type = 0;
                (void)hipMemcpy(&type, d_raw + ends[bin], 8, hipMemcpyDeviceToHost);
                ctx->err = "Invalid bin type, unknown value: " + std::to_string((long long)type);  // BlockManager.scala:85-87
                break;
            }
            case kBlkCount: ctx->err = "Encoded position count should be greater than zero"; break;  // :232-233
            default: ctx->err = "Failed to correctly parse block, the number of position entries exceeds the buffer size"; break;  // :235-236
        }
        return FFH_E_FORMAT;
    };
    FFH_HIP(hipMemcpyAsync(&n_payload, d_pbase.p + nb, 8, hipMemcpyDeviceToHost, st));
    FFH_HIP(hipStreamSynchronize(st));  // bins with a bad header have no payload; the walk still visits the others so that the FIRST bad bin is reported
    if (n_payload >= (1ull << 32) - 64) { ctx->err = "more than 2^32 payload longs in one shard; split the bins across more GPUs"; return FFH_E_ARG; }
    FFH_HIP(d_marks.reserve(n_payload + 8));
    FFH_HIP(d_rank.reserve(n_payload + 8));
    FFH_HIP(d_scr32.reserve(scan_scratch_elems_safe(n_payload + 1)));
    FFH_HIP(hipMemsetAsync(d_marks.p, 0, (size_t)(n_payload + 8) * 4, st));
    if (nb) hipLaunchKernelGGL(k_block_walk, dim3(nb), dim3(256), 0, st, d_raw, d_off.p, nb, d_hdr.p, d_plen.p, d_pbase.p, d_marks.p, d_err);
    exclusive_scan<uint32_t, uint32_t>(d_marks.p, n_payload, d_rank.p, d_scr32.p, st);
    uint32_t nt = 0;
    FFH_HIP(hipMemcpyAsync(&herr, d_err, 8, hipMemcpyDeviceToHost, st));
    FFH_HIP(hipMemcpyAsync(&nt, d_rank.p + n_payload, 4, hipMemcpyDeviceToHost, st));
    FFH_HIP(hipStreamSynchronize(st));
    if (herr != ~0ull) return report(herr);
    ctx->T = nt;
    ctx->P = n_payload - nt;
    FFH_HIP(ctx->targets.reserve(ctx->T + 1));
    FFH_HIP(ctx->positions.reserve(ctx->P + 1));
    if (n_payload)
        hipLaunchKernelGGL(k_block_split, dim3(blocks_for(n_payload, 256)), dim3(256), 0, st, d_raw, d_off.p, nb, d_hdr.p, d_pbase.p, d_rank.p, n_payload,
                           ctx->targets.p, ctx->positions.p);
    FFH_HIP(hipGetLastError());
    FFH_HIP(hipStreamSynchronize(st));
    return FFH_OK;
}

int ffh_db_load_blocks(ffh_ctx *ctx, const int64_t *longs, const uint64_t *bin_offsets, uint32_t n_bins) {
    if (!ctx || !longs || !bin_offsets) { if (ctx) ctx->err = "null argument"; return FFH_E_ARG; }
    if (ctx->enzyme == 0) { ctx->err = "the context has no enzyme yet: create it with an enzyme index or open a database file"; return FFH_E_STATE; }
    FFH_HIP(hipSetDevice(ctx->device));
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> off(bin_offsets, bin_offsets + n_bins), len(n_bins);
    for (uint32_t b = 0; b < n_bins; ++b) {
        if (bin_offsets[b + 1] < bin_offsets[b]) { ctx->err = "bin offsets must not decrease"; return FFH_E_ARG; }
        len[b] = bin_offsets[b + 1] - bin_offsets[b];
        off[b] -= bin_offsets[0];
    }
    const uint64_t n_longs = n_bins ? bin_offsets[n_bins] - bin_offsets[0] : 0;
    DevBuf<int64_t> d_raw;
    FFH_HIP(d_raw.reserve(n_longs + 1));
    if (n_longs) FFH_HIP(hipMemcpyAsync(d_raw.p, longs + bin_offsets[0], n_longs * 8, hipMemcpyHostToDevice, ctx->st));
    const auto t1 = std::chrono::steady_clock::now();
    int rc = decode_blocks_on_device(ctx, d_raw.p, off, len);
    d_raw.release();
    if (rc) return rc;
    ctx->load = ffh_load_stats{};
    ctx->load.raw_bytes = n_longs * 8;
    ctx->load.inflate_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    ctx->load.decode_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    ctx->n_bins = n_bins; ctx->bin_begin = 0; ctx->bin_end = n_bins;
    return prepare_database(ctx);
}

int ffh_db_open(ffh_ctx *ctx, const char *db_path, uint32_t bin_begin, uint32_t bin_end) {
    if (!ctx || !db_path) { if (ctx) ctx->err = "null argument"; return FFH_E_ARG; }
    const auto t0 = std::chrono::steady_clock::now();
    DbHeader h;
    std::string e = read_db_header(std::string(db_path) + ".header", h);
    if (!e.empty()) { ctx->err = e; return e.rfind("cannot open", 0) == 0 ? FFH_E_IO : FFH_E_FORMAT; }
    if (ctx->enzyme == 0) set_enzyme(ctx, h.enzyme_index);
    if (h.enzyme_index != ctx->enzyme) {
        // the context was created for another enzyme than the one recorded in the header (BinaryHeader.scala:127)
        ctx->err = "database enzyme index " + std::to_string(h.enzyme_index) + " differs from the context's " + std::to_string(ctx->enzyme);
        return FFH_E_ARG;
    }
    if (bin_end == 0 || bin_end > h.n_bins) bin_end = h.n_bins;
    FFH_HIP(hipSetDevice(ctx->device));
    BodyFile body;
    std::vector<uint64_t> off, len;
    uint64_t need_lo = 0, need_hi = 0;
    e = open_body(db_path, body);
    if (e.empty()) e = locate_bins(body, h, bin_begin, bin_end, need_lo, need_hi, off, len);
    if (!e.empty()) { ctx->err = e; return e.rfind("cannot open", 0) == 0 ? FFH_E_IO : FFH_E_FORMAT; }
    const auto t1 = std::chrono::steady_clock::now();
    DevBuf<int64_t> d_raw;
    IngestStats is;
    const int64_t *payload = nullptr;  // first long of bin_begin's payload
    size_t m0 = 0, m1 = 0;
    member_range(body, need_lo, need_hi, m0, m1);
    bool on_device = !ctx->sw.inflate_host && m1 > m0;   // (FFH_INFLATE=host: inflate on the host threads instead of on the device)
    if (on_device && (need_lo - body.members[m0].uoff) % 8) on_device = false;  // the payload must stay 8-byte aligned inside the members' output
    ctx->load_device_inflate_ms = 0;
    if (on_device) {
        const Member &first = body.members[m0], &last = body.members[m1 - 1];
        const uint64_t ubase = first.uoff, uspan = last.uoff + last.isize - ubase;
        const size_t cspan = last.cdata_off + last.cdata_len + 8 - first.coff;
        DevBuf<uint8_t> d_comp;
        DevBuf<InflateMember> d_mem;
        DevBuf<uint16_t> d_work;
        DevBuf<uint32_t> d_crc;
        FFH_HIP(d_comp.reserve(cspan + 64));
        FFH_HIP(d_raw.reserve(uspan / 8 + 2));
        e = inflate_to_device(body, need_lo, need_hi, d_comp.p, ctx->device, is, true);
        if (!e.empty()) { d_raw.release(); ctx->err = e; return FFH_E_FORMAT; }
        const auto td = std::chrono::steady_clock::now();
        std::vector<InflateMember> hm;
        hm.reserve(m1 - m0);
        for (size_t i = m0; i < m1; ++i) {
            const Member &m = body.members[i];
            if (m.isize == 0) continue;
            hm.push_back(InflateMember{m.cdata_off - first.coff, m.uoff - ubase, (uint32_t)m.cdata_len, m.isize, m.crc, 0u});
        }
        std::vector<uint32_t> crc_tab(8 * 256);
        for (uint32_t v = 0; v < 256; ++v) {
            uint32_t c = v;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            crc_tab[v] = c;
        }
        for (int k = 1; k < 8; ++k)
            for (uint32_t v = 0; v < 256; ++v) crc_tab[(size_t)k * 256 + v] = (crc_tab[(size_t)(k - 1) * 256 + v] >> 8) ^ crc_tab[crc_tab[(size_t)(k - 1) * 256 + v] & 0xFF];
        const uint32_t nm = (uint32_t)hm.size(), batch = std::min<uint32_t>(nm, 1u << 18);  // one launch up to 4x hg38: the kernel is latency-bound per member
        FFH_HIP(d_mem.reserve(nm + 1));
        FFH_HIP(d_work.reserve((size_t)batch * kInflateWorkU16 + 64));
        FFH_HIP(d_crc.reserve(8 * 256));
        unsigned long long *d_err = ctx->d_counters + 11;
        FFH_HIP(hipMemcpyAsync(d_mem.p, hm.data(), (size_t)nm * sizeof(InflateMember), hipMemcpyHostToDevice, ctx->st));
        FFH_HIP(hipMemcpyAsync(d_crc.p, crc_tab.data(), crc_tab.size() * 4, hipMemcpyHostToDevice, ctx->st));
        FFH_HIP(hipMemsetAsync(d_err, 0xFF, 8, ctx->st));
        for (uint32_t b0 = 0; b0 < nm; b0 += batch) {
            const uint32_t nb = std::min(batch, nm - b0);
            hipLaunchKernelGGL(k_inflate<64>, dim3(blocks_for(nb, 64)), dim3(64), 0, ctx->st, (const uint8_t *)d_comp.p, (const InflateMember *)d_mem.p, b0, nb,
                               (uint8_t *)d_raw.p, d_work.p, d_err);
        }
        if (nm) hipLaunchKernelGGL(k_crc32, dim3(blocks_for(nm, 64)), dim3(64), 0, ctx->st, (const uint8_t *)d_raw.p, (const InflateMember *)d_mem.p, nm,
                                   (const uint32_t *)d_crc.p, d_err);
        unsigned long long herr = ~0ull;
        FFH_HIP(hipMemcpyAsync(&herr, d_err, 8, hipMemcpyDeviceToHost, ctx->st));
        FFH_HIP(hipStreamSynchronize(ctx->st));
        if (herr != ~0ull) {
            d_raw.release();
            ctx->err = "BGZF inflate / crc failure in the database body (member " + std::to_string((unsigned long long)(herr >> 8)) + ", code " + std::to_string((unsigned)(herr & 0xFF)) + ")";
            return FFH_E_FORMAT;
        }
        ctx->load_device_inflate_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td).count();
        payload = d_raw.p + (need_lo - ubase) / 8;
    } else {
        FFH_HIP(d_raw.reserve((need_hi - need_lo) / 8 + 1));
        e = inflate_to_device(body, need_lo, need_hi, (uint8_t *)d_raw.p, ctx->device, is, false);
        if (!e.empty()) { d_raw.release(); ctx->err = e; return FFH_E_FORMAT; }
        payload = d_raw.p;
    }
    const auto t2 = std::chrono::steady_clock::now();
    int rc = decode_blocks_on_device(ctx, payload, off, len);
    d_raw.release();
    if (rc) return rc;
    const auto t3 = std::chrono::steady_clock::now();
    ctx->contigs = h.contigs;
    ctx->bin_bytes = h.uncompressed_bytes;
    ctx->load = ffh_load_stats{};
    ctx->load.open_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    ctx->load.inflate_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    ctx->load.decode_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
    ctx->load.compressed_bytes = is.compressed_bytes;
    ctx->load.raw_bytes = is.raw_bytes;
    ctx->load.threads = is.threads;
    ctx->load.device_inflate_ms = ctx->load_device_inflate_ms;
    rc = prepare_database(ctx);
    ctx->n_bins = h.n_bins; ctx->bin_begin = bin_begin; ctx->bin_end = bin_end;
    return rc;
}

int ffh_db_load_stats(const ffh_ctx *ctx, ffh_load_stats *out) {
    if (!ctx || !out) return FFH_E_ARG;
    *out = ctx->load;
    out->prepare_ms = ctx->db_prepare_ms;
    return FFH_OK;
}

int ffh_db_open_header(ffh_ctx *ctx, const char *db_path) {
    if (!ctx || !db_path) { if (ctx) ctx->err = "null argument"; return FFH_E_ARG; }
    DbHeader h;
    const std::string e = read_db_header(std::string(db_path) + ".header", h);
    if (!e.empty()) { ctx->err = e; return e.rfind("cannot open", 0) == 0 ? FFH_E_IO : FFH_E_FORMAT; }
    if (ctx->enzyme == 0) set_enzyme(ctx, h.enzyme_index);
    if (h.enzyme_index != ctx->enzyme) { ctx->err = "database enzyme index differs from the context's"; return FFH_E_ARG; }
    ctx->contigs = h.contigs;
    ctx->bin_bytes = h.uncompressed_bytes;
    ctx->n_bins = h.n_bins;
    return FFH_OK;
}

uint64_t ffh_db_bin_bytes(const ffh_ctx *ctx, uint32_t bin) { return (ctx && bin < ctx->bin_bytes.size()) ? ctx->bin_bytes[bin] : 0; }

int ffh_db_info_get(const ffh_ctx *ctx, ffh_db_info *out) {
    if (!ctx || !out) return FFH_E_ARG;
    out->n_targets = ctx->T; out->n_positions = ctx->P; out->n_bins = ctx->n_bins; out->bin_begin = ctx->bin_begin; out->bin_end = ctx->bin_end;
    out->enzyme_index = ctx->enzyme; out->prefix_bases = ctx->img[0].width; out->suffix_bases = ctx->img[1].width; out->prepare_ms = ctx->db_prepare_ms;
    return FFH_OK;
}

const char *ffh_db_contig(const ffh_ctx *ctx, uint32_t id) {
    if (!ctx || id < 1 || id > ctx->contigs.size()) return nullptr;
    return ctx->contigs[id - 1].c_str();
}

// ---- slabs of a bounded scan -------------------------------------------------------------------------------------
// The database in database order is cut at prefix-bucket boundaries into slabs of growing size (1/64, 1/8, the rest); the prefix
// image serves every slab through a bucket range, the suffix image exists once per slab.  Needs database order == prefix-bucket
// order, i.e. a 3' PAM (every Cas9 pack); Cpf1's 5' PAM varies in front of the compared bases, so a Cpf1 context stays unbounded.
// sorted by sequence?  and the first target of every first-three-bases rank r (cut[r] = smallest index with rank >= r)
__global__ void k_slab_cuts(const uint64_t *__restrict__ targets, uint64_t n, int scan_len, uint32_t *__restrict__ bad, uint32_t *__restrict__ cut /* [65], preset to n */) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t M = (1ull << 48) - 1ull, t = targets[i] & M;
    const uint32_t r = (uint32_t)(t >> (2 * scan_len - 6)) & 63u;
    if (i == 0) { for (uint32_t k = 0; k <= r; ++k) cut[k] = 0u; return; }
    const uint64_t p = targets[i - 1] & M;
    const uint32_t rp = (uint32_t)(p >> (2 * scan_len - 6)) & 63u;
    if (p > t || rp > r) atomicAdd(bad, 1u);   // (rp > r: bits above the sequence are set and lead the order -- not a database the reference writes)
    for (uint32_t k = rp + 1; k <= r; ++k) cut[k] = (uint32_t)i;   // ranks without a target between rp and r start here too
}

static void drop_slabs(ffh_ctx *ctx) { ctx->slab_img.clear(); ctx->slab_t.clear(); ctx->slabs_state = 0; }

// (round 3 tried eight slabs whose ends double -- 1/64, 1/32, ... 1/2, 3/4, 1 -- once a slab no longer re-enumerated the prefix
// candidates: the same 4.8e7 raw hits on the repeat-structured workload -- the guides that overshoot do so inside the FIRST slab, a
// 64th of a million-copy family is 15 000 hits -- and 1.1 ms more in two more compare launches; six slabs stay)
// slab k = the targets whose first three bases rank in [kSlabRank[k], kSlabRank[k + 1]): 1/64, 3/64, 1/8, 3/16, 1/4 and 3/8 of
// sequence space.  A guide with H hits spread like the genome is retired after the first slab boundary beyond 2000/H of it, so it
// leaves at most ~1.6 x the hits its cut-off keeps plus one slab's worth; a guide of a million-copy family leaves 1/64 of them.
// On the repeat-structured bench workload (2.5e8 raw hits unbounded): 3 slabs {1, 8} 1.43e8 raw hits / 16.5 ms per step,
// 4 slabs {1, 8, 32} 7.0e7 / 15.2 ms, these 6 slabs 4.7e7 / 13.7 ms (19.1 ms unbounded); every slab costs ~0.8 ms of its own
// (candidate binning with a counting pass, ordering and totals of its hits, two round trips).
static const uint32_t kSlabRank[7] = {0u, 1u, 4u, 12u, 24u, 40u, 64u};

static int ensure_slabs(ffh_ctx *ctx) {
    if (ctx->slabs_state) return FFH_OK;   // 1 = built, -1 = this database cannot be bounded
    ctx->slabs_state = -1;
    const int sfx = ctx->img[1].width;
    if (ctx->geo.c0 == 0 || ctx->T < (1u << 16) || ctx->img[0].width < 3) return FFH_OK;
    hipStream_t st = ctx->st;
    uint32_t *bad = (uint32_t *)ctx->d_counters + 9;
    DevBuf<uint32_t> d_cut;
    FFH_HIP(d_cut.reserve(66));
    std::vector<uint32_t> cut(65, (uint32_t)ctx->T);
    FFH_HIP(hipMemsetAsync(bad, 0, 4, st));
    FFH_HIP(hipMemcpyAsync(d_cut.p, cut.data(), 65 * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_slab_cuts, dim3(blocks_for(ctx->T, 256)), dim3(256), 0, st, (const uint64_t *)ctx->targets.p, ctx->T, ctx->geo.scan_len, bad, d_cut.p);
    uint32_t hbad = 1;
    FFH_HIP(hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, st));
    FFH_HIP(hipMemcpyAsync(cut.data(), d_cut.p, 65 * 4, hipMemcpyDeviceToHost, st));
    FFH_HIP(hipStreamSynchronize(st));
    if (hbad) return FFH_OK;   // (targets not in sequence order: ffh_db_load_soa takes what it is given)
    const int K = (int)(sizeof kSlabRank / sizeof kSlabRank[0]) - 1;
    for (int k = 0; k < K; ++k) {
        const uint64_t t0 = cut[kSlabRank[k]], t1 = kSlabRank[k + 1] >= 64u ? ctx->T : cut[kSlabRank[k + 1]];
        ctx->slab_img.emplace_back(new Image());
        const int rc = build_image_into(ctx, *ctx->slab_img.back(), 1, sfx, t0, t1 - t0);
        if (rc) { drop_slabs(ctx); ctx->slabs_state = -1; return rc; }
        ctx->slab_t.push_back(t0);
    }
    ctx->slab_t.push_back(ctx->T);
    FFH_HIP(hipStreamSynchronize(st));
    ctx->tmp_keys.release(); ctx->tmp_tidx.release();
    ctx->slabs_state = 1;
    return FFH_OK;
}

// The split of the compared bases into prefix and suffix key is fixed when the images are built; the best split depends on
// maxMismatch (at hg38 scale 11 + 9 for <= 4 mismatches, 10 + 10 for 5: 40 % fewer pair tests).  A large database therefore keeps
// up to two pairs of images: when the cost model prefers another width by more than a fifth, that pair is built once (tens of ms)
// and the two pairs are swapped per call.  Off for forced plans (ffh_set_plan) and small databases.
static int select_images(ffh_ctx *ctx, int max_mm) {
    if (!ctx->auto_width || ctx->plan_a >= 0 || ctx->plan_r1 >= 0 || ctx->T < (1ull << 24)) return FFH_OK;
    const int lc = ctx->geo.lc, cur = ctx->img[0].width;
    const double T = (double)ctx->T;
    Plan p;
    const double cost_cur = plan_cost(T, ctx->span, lc, cur, max_mm, p);
    // among equally cheap widths (the model is symmetric in prefix and suffix) the one nearest to the default split wins
    const int a_def = default_prefix_width(T, ctx->span, lc);
    int best_a = cur;
    double best = cost_cur;
    for (int a = std::max(lc - 12, 8); a <= std::min(12, lc - 8); ++a) {
        const double c = plan_cost(T, ctx->span, lc, a, max_mm, p);
        if (c < best * (1.0 - 1e-9) || (c <= best * (1.0 + 1e-9) && std::abs(a - a_def) < std::abs(best_a - a_def))) { best = c; best_a = a; }
    }
    if (best_a == cur || best > 0.8 * cost_cur) return FFH_OK;
    if (ctx->alt[0].width != best_a) {
        int rc = build_image_into(ctx, ctx->alt[0], 0, best_a, 0, ctx->T);
        if (!rc) rc = build_image_into(ctx, ctx->alt[1], 1, lc - best_a, 0, ctx->T);
        FFH_HIP(hipStreamSynchronize(ctx->st));
        ctx->tmp_keys.release(); ctx->tmp_tidx.release();
        if (rc) { ctx->alt[0] = Image(); ctx->alt[1] = Image(); return rc; }
    }
    std::swap(ctx->img[0], ctx->alt[0]);
    std::swap(ctx->img[1], ctx->alt[1]);
    drop_slabs(ctx);   // (the slabs' suffix images have the other width)
    return FFH_OK;
}

// ---- order n keys (guide << tbits | database index; `n_real` of them hits, the rest all-ones chunk padding) by (guide, index) and leave
// every guide's segment in seg_begin / seg_end (cleared by the caller; indices relative to the ordered array) ----
// keys = base of the n records; alt_buf + alt_off = a scratch range of the same length.  *sorted = where the ordered records are (keys
// or the scratch range), *n_out = how many there are (the bin path drops the padding, the others sort it behind the hits).
//   up to 4096 records          one block, bitonic network in LDS
//   a moderate number of hits   (round 5) one unstable most-significant-digit pass into 2^B bins of a few thousand + one launch that orders
//                               every bin inside LDS and leaves the segment bounds (ffh_prims.hpp: k_msd_*, k_binsort)
//   > 256 hits per guide        the device-wide LSD sort over all key bits: segments of thousands of hits -- a 5-mismatch scan, guides
//                               inside repeat families -- are what the device-wide passes are good at
//   otherwise                   two device-wide passes over the guide bits, then one wave per guide orders its segment (k_segsort; guides
//                               inside repeat families go to k_segsort_heavy)
// FFH_SORT=lsd / seg / bin forces one (A/B runs, tests).
static int order_hits(ffh_ctx *ctx, hipStream_t st, uint64_t *keys, DevBuf<uint64_t> &alt_buf, uint64_t alt_off, uint64_t n, uint64_t n_real, int gbits, uint32_t n_guides,
                      uint32_t *seg_begin, uint32_t *seg_end, uint64_t **sorted, uint64_t *n_out) {
    *sorted = keys; *n_out = n;
    if (!n) return FFH_OK;
    const bool full_lsd = ctx->sw.sort_mode == 1, force_seg = ctx->sw.sort_mode == 2, force_bin = ctx->sw.sort_mode == 3;
    bool segments_done = false;
    if (n <= kBinCap && gbits <= kBinMaxSubBits && !full_lsd && !force_seg) {
        // a small scan (a chr22-scale call: a few thousand records, <= 2048 guides): k_binsort alone, one block, in place -- the records
        // counted per guide in LDS, every guide's indices ordered by a wave, segment bounds left behind; the chunk padding is dropped
        // (round 4: a 4096-key bitonic network in one block, 52 us, + k_segments)
        hipLaunchKernelGGL(k_binsort, dim3(1), dim3(kMsdThreads), 0, st, keys, (const uint32_t *)nullptr, 0u, 1u, n, ctx->tbits, gbits, n_guides, seg_begin, seg_end,
                           (uint32_t *)nullptr, (uint32_t *)nullptr);
        *n_out = n_real;
        segments_done = true;
    } else if (n <= kSmallSort) hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(1024), 0, st, keys, (uint32_t)n);
    else {
        const uint32_t nbk = sort_nblocks(n);
        FFH_HIP(alt_buf.reserve(std::max<size_t>(ctx->hits.cap, (size_t)(alt_off + n))));
        uint64_t *alt = alt_buf.p + alt_off;
        FFH_HIP(ctx->sort_table.reserve((size_t)kSortTableDigits * nbk + 1));
        FFH_HIP(ctx->sort_offs.reserve((size_t)kSortTableDigits * nbk + 1));
        FFH_HIP(ctx->scan_tmp32.reserve(scan_scratch_elems_safe((uint64_t)kSortTableDigits * nbk)));
        SortScratch ss;
        ss.alt = alt; ss.table = ctx->sort_table.p; ss.offs = ctx->sort_offs.p; ss.scan_tmp = ctx->scan_tmp32.p;
        const bool many = n > 256ull * std::max<uint32_t>(n_guides, 1u);
        int B = 0;
        while (B < std::min(gbits, (int)kMsdMaxBits) && (n >> B) > 4096) ++B;
        const int sub_bits = gbits - B;
        const bool bins_fit = sub_bits <= kBinMaxSubBits && (double)n / (double)(1u << B) <= 0.65 * kBinCap;
        uint32_t *n_heavy = (uint32_t *)(ctx->d_counters + 13);   // (cleared by k_compare_setup: every call here follows a compare launch)
        if (bins_fit && !full_lsd && !force_seg && (!many || force_bin)) {
            const uint32_t nbins = 1u << B, nbm = msd_nblocks(n);
            const int shift = ctx->tbits + gbits - B;
            FFH_HIP(ctx->sort_table.reserve((size_t)nbins * nbm + 1));
            FFH_HIP(ctx->sort_offs.reserve((size_t)nbins * nbm + 1));
            FFH_HIP(ctx->scan_tmp32.reserve(scan_scratch_elems_safe((uint64_t)nbins * nbm)));
            FFH_HIP(ctx->heavy_list.reserve((size_t)std::max<uint32_t>(n_guides, nbins) + 1));
            hipLaunchKernelGGL(k_msd_hist, dim3(nbm), dim3(kMsdThreads), 0, st, (const uint64_t *)keys, n, shift, nbins, ctx->tbits, n_guides, ctx->sort_table.p, nbm);
            exclusive_scan<uint32_t, uint32_t>(ctx->sort_table.p, (uint64_t)nbins * nbm, ctx->sort_offs.p, ctx->scan_tmp32.p, st);
            hipLaunchKernelGGL(k_msd_scatter, dim3(nbm), dim3(kMsdThreads), 0, st, (const uint64_t *)keys, alt, n, shift, nbins, ctx->tbits, n_guides, (const uint32_t *)ctx->sort_offs.p, nbm);
            // (the scatter dropped the chunk padding: from here on the array holds the n_real hits, contiguously)
            hipLaunchKernelGGL(k_binsort, dim3(nbins), dim3(kMsdThreads), 0, st, alt, (const uint32_t *)ctx->sort_offs.p, nbm, nbins, n_real, ctx->tbits, sub_bits, n_guides, seg_begin,
                               seg_end, ctx->heavy_list.p, n_heavy);
            hipLaunchKernelGGL(k_binsort_heavy, dim3(256), dim3(256), 0, st, alt, keys, (const uint32_t *)ctx->sort_offs.p, nbm, nbins, n_real, (const uint32_t *)ctx->heavy_list.p,
                               (const uint32_t *)n_heavy, ctx->tbits, sub_bits, n_guides, seg_begin, seg_end);
            *sorted = alt; *n_out = n_real;
            segments_done = true;
        } else if (full_lsd || (many && !force_seg)) *sorted = radix_sort_u64(keys, n, 0, ctx->tbits + gbits, 64, 64, ss, st);
        else {
            FFH_HIP(ctx->heavy_list.reserve((size_t)n_guides + 1));
            uint64_t *by_guide = radix_sort_u64(keys, n, ctx->tbits, ctx->tbits + gbits, 64, 64, ss, st);
            uint64_t *other = by_guide == keys ? alt : keys;
            hipLaunchKernelGGL(k_segments, dim3(blocks_for(n, 256)), dim3(256), 0, st, by_guide, n, ctx->tbits, n_guides, seg_begin, seg_end);
            hipLaunchKernelGGL(k_segsort, dim3(blocks_for(n_guides, 4)), dim3(256), 0, st, by_guide, (const uint32_t *)seg_begin, (const uint32_t *)seg_end, n_guides, ctx->tbits,
                               ctx->heavy_list.p, n_heavy);
            hipLaunchKernelGGL(k_segsort_heavy, dim3(512), dim3(256), 0, st, by_guide, other, (const uint32_t *)seg_begin, (const uint32_t *)seg_end, (const uint32_t *)ctx->heavy_list.p,
                               (const uint32_t *)n_heavy, ctx->tbits);
            *sorted = by_guide;
            segments_done = true;
        }
    }
    if (!segments_done) hipLaunchKernelGGL(k_segments, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const uint64_t *)*sorted, n, ctx->tbits, n_guides, seg_begin, seg_end);
    FFH_HIP(hipGetLastError());
    return FFH_OK;
}

// Address and capacity of every device buffer the candidate-list / work-list launches of a scan read or write (prepare_side, side_plan
// and the SideArgs they fill), folded into one word: a captured sequence is replayed only while this is what it was when the sequence
// was captured.  Per context: another context's allocations (another shard's thread, a finalize buffer that grows) do not touch it.
static uint64_t prep_signature(const ffh_ctx *ctx, const Image &suffix) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ctx->db_gen;
    auto mix = [&](const void *p, size_t cap) {
        h ^= (uint64_t)(uintptr_t)p + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h ^= (uint64_t)cap + 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    };
    mix(ctx->guides.p, ctx->guides.cap); mix(ctx->seg_begin.p, ctx->seg_begin.cap); mix(ctx->seg_end.p, ctx->seg_end.cap);
    mix(ctx->item_gid.p, ctx->item_gid.cap);
    for (int w = 0; w < 2; ++w) {
        mix(ctx->gtab[w].p, ctx->gtab[w].cap); mix(ctx->gbucket[w].p, ctx->gbucket[w].cap); mix(ctx->patterns[w].p, ctx->patterns[w].cap);
        mix(ctx->istart[w].p, ctx->istart[w].cap); mix(ctx->part_pairs[w].p, ctx->part_pairs[w].cap);
        const ffh_ctx::SideScratch &sc = ctx->side_scr[w];
        mix(sc.part_fill.p, sc.part_fill.cap); mix(sc.part_hist.p, sc.part_hist.cap); mix(sc.part_start.p, sc.part_start.cap);
        mix(sc.gp_start.p, sc.gp_start.cap); mix(sc.by_part.p, sc.by_part.cap); mix(sc.scan_tmp.p, sc.scan_tmp.cap);
        mix(ctx->wl_count[w].p, ctx->wl_count[w].cap); mix(ctx->wl_list[w].p, ctx->wl_list[w].cap);
        const Image &im = w == 0 ? ctx->img[0] : suffix;
        mix(im.bstart.p, im.bstart.cap); mix(im.gstart.p, im.gstart.cap); mix(im.gwords.p, im.gwords.cap); mix(im.tidx.p, im.tidx.cap); mix(im.live.p, im.live.cap);
    }
    return h | 1ull;   // (never 0: the value of "nothing seen yet")
}

// bound_ot > 0: the caller will not ask for more than bound_ot positions per guide (maximumOffTargets), so a guide whose positions
// reach it in the slabs scanned so far is retired from the later ones
static int scan_impl(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mm, uint32_t bound_ot) {
    if (ctx) ctx->too_many_hits = false;
    if (!ctx || (n_guides && !guides) || max_mm < 0) { if (ctx) ctx->err = "bad argument"; return FFH_E_ARG; }
    if (ctx->img[0].width < 0) { ctx->err = "no database loaded"; return FFH_E_STATE; }
    FFH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->st;
    ctx->scanned = false;
    ctx->n_guides = n_guides;
    ctx->max_mm = max_mm;
    ctx->tm = ffh_timings{};
    FFH_HIP(ctx->guides.reserve((size_t)n_guides + 1));
    FFH_HIP(ctx->seg_begin.reserve((size_t)n_guides + 1));  // cleared per batch by k_guide_keys, filled by k_segments
    FFH_HIP(ctx->seg_end.reserve((size_t)n_guides + 1));
    // host or device memory (unified addressing tells): a caller whose guide set already sits in HBM passes the device pointer
    if (n_guides && guides != ctx->guides.p) FFH_HIP(hipMemcpyAsync(ctx->guides.p, guides, (size_t)n_guides * 8, hipMemcpyDefault, st));
    if (ctx->hits.cap == 0) FFH_HIP(ctx->hits.reserve(std::max<size_t>(1u << 22, (size_t)n_guides * 256)));

    { const int rc = select_images(ctx, std::min(max_mm, ctx->geo.lc)); if (rc) return rc; }
    ctx->tbits = 1;
    while (ctx->tbits < 32 && (1ull << ctx->tbits) < std::max<uint64_t>(ctx->T, 2)) ++ctx->tbits;
    int gbits = 1;   // 2^gbits > n_guides: the all-ones padding of the compare waves' chunks sorts behind every guide
    while (gbits < 32 && (1ull << gbits) <= (uint64_t)n_guides) ++gbits;
    const Plan plan = choose_plan(ctx, std::min(max_mm, ctx->geo.lc));
    ctx->tm.prefix_bases = plan.a; ctx->tm.prefix_radius = plan.r1; ctx->tm.suffix_radius = plan.r2;
    const double np_p = ball_size(plan.a, plan.r1), np_s = ball_size(plan.s, plan.r2);
    // batch size: the candidate CSR of both images must stay addressable with 32 bits (and a sane size)
    double max_batch = (double)std::max<uint32_t>(n_guides, 1);
    max_batch = std::min(max_batch, (double)(1ull << 30) / (np_p + np_s));
    max_batch = std::min(max_batch, (double)((1u << kGidBits) - 1u));
    if (ctx->max_guide_batch) max_batch = std::min(max_batch, (double)ctx->max_guide_batch);
    uint32_t batch = (uint32_t)std::max(1.0, std::floor(max_batch));
    // the slabs of this scan: one (everything) unless the scan is bounded and the database allows it
    struct Slab { const Image *suffix; uint32_t rank_lo, rank_hi; uint64_t n_targets; };
    std::vector<Slab> slabs;
    bool bounded = bound_ot > 0 && ctx->bound_mode > 0 && n_guides > 0 && plan.r2 >= 0;
    // (slab images that cannot be had -- out of memory for the six extra suffix images -- mean an unbounded scan, not a failed one)
    if (bounded) { if (ensure_slabs(ctx) != FFH_OK) { ctx->err.clear(); (void)hipGetLastError(); } bounded = ctx->slabs_state == 1; }
    if (bounded)
        for (size_t k = 0; k + 1 < ctx->slab_t.size(); ++k)
            slabs.push_back(Slab{ctx->slab_img[k].get(), kSlabRank[k], kSlabRank[k + 1] - 1u, ctx->slab_t[k + 1] - ctx->slab_t[k]});
    else slabs.push_back(Slab{&ctx->img[1], 0u, 63u, ctx->T});
    ctx->bound_ot = bounded ? bound_ot : 0u;
    ctx->tm.bounded_slabs = bounded ? (uint32_t)slabs.size() : 0u;
    // How each image is cut into work entries for the compare kernel (ffh_compare.hpp): runs of NB small buckets sized so that a
    // typical run fills ~3/4 of the wave's LDS strip (kKW words of groups, kKC candidates); k_work_count / k_work_fill then list the
    // runs that have candidates, a bucket larger than the strip as several strip-sized group ranges.  Candidate lists larger than the
    // strip take the kernel's piecewise path.
    double expect[2] = {0.0, 0.0};   // work entries the two lists are expected to hold (the compare launch's way of dealing them depends on it)
    auto side_plan = [&](hipStream_t st, int which, const Image &im, uint64_t n_targets, int width, int r_far, double n_patterns, uint32_t ng, SideArgs &S,
                         uint32_t rank_lo = 0u, uint32_t rank_hi = 63u, bool count_pairs = true) -> int {
        S = SideArgs{};
        S.gstart = im.gstart.p; S.gwords = im.gwords.p; S.tidx = im.direct ? nullptr : im.tidx.p; S.dd_off = im.direct ? S_nb_plus_1(width) : 0u; S.istart = ctx->istart[which].p; S.gtab = ctx->gtab[which].p;
        S.nb = 1u << (2 * width); S.width = (uint32_t)width; S.rest = (uint32_t)im.rest; S.r_far = r_far;
        const double cap_g = std::floor((double)kKW / group_words(im.rest));
        const double avg_t = (double)n_targets / (double)S.nb, avg_g = avg_t / 32.0 + (avg_t > 0 ? 0.5 : 0.0), avg_c = (double)ng * n_patterns / (double)S.nb;
        // (round 5: 0.85 / 0.8 of the strip instead of 0.75 / 0.7 -- 15 instead of 13 prefix buckets per entry at hg38 scale: fuller rows and
        // 13 % fewer entries to park, 1.000 against 1.026 ms per launch, profiles/r05/ab_log.txt 1)
        const double by_groups = std::floor(0.85 * cap_g / std::max(avg_g, 0.25)), by_cands = std::floor(0.8 * kKC / std::max(avg_c, 0.05));
        S.NB = (uint32_t)std::max(1.0, std::min((double)kMaxNB, std::min(by_groups, by_cands)));
        if (ctx->sw.nb_force[which] > 0) S.NB = (uint32_t)std::min(ctx->sw.nb_force[which], kMaxNB);
        S.split = (uint32_t)cap_g;
        const uint32_t n_bat = (S.nb + S.NB - 1) / S.NB;
        {   // batches that have a target and a candidate: the shard's part of prefix-key space (plan_cost), the slab's ranks
            const double part = (which == 0 ? ctx->span : 1.0) * (double)(rank_hi - rank_lo + 1u) / 64.0;
            expect[which] = std::min({(double)n_bat * part, (double)ng * n_patterns * part, (double)n_targets * part});
        }
        // (what a guide set without pile-ups needs; one that needs more is noticed after the launch, which then runs again: below)
        const uint64_t max_entries = (uint64_t)n_bat + 2 * ((n_targets / 32 + S.nb) / S.split + 1) + (uint64_t)((double)ng * n_patterns) / kKC + 2;
        FFH_HIP(ctx->wl_list[which].reserve(ctx->sw.work_list_limit > 0 ? std::min<size_t>((size_t)max_entries, (size_t)ctx->sw.work_list_limit) : (size_t)max_entries));
        S.list_cap = (uint32_t)std::min<size_t>(ctx->wl_list[which].cap, 0xFFFFFFF0u);
        // (FFH_WORK_LIST_LIMIT: test aid -- a first list that small, so that the run-again path below is taken)
        const unsigned wblocks = blocks_for(n_bat, kWorkThreads);
        FFH_HIP(ctx->wl_count[which].reserve((size_t)n_bat + wblocks + 64));   // [n_bat counts][wblocks block sums][.. the list's length]
        uint32_t *counts = ctx->wl_count[which].p, *block_sums = counts + n_bat;
        hipLaunchKernelGGL(k_work_count, dim3(wblocks), dim3(kWorkThreads), 0, st, im.gstart.p, ctx->istart[which].p, S.nb, S.NB, S.split, n_bat, counts, block_sums,
                           (const unsigned long long *)ctx->part_pairs[which].p, count_pairs ? ctx->n_part[which] : 0u, ctx->d_counters + kStatPairs + which, rank_lo, rank_hi,
                           (uint32_t)width);
        hipLaunchKernelGGL(k_work_fill, dim3(wblocks), dim3(kWorkThreads), 0, st, im.gstart.p, ctx->istart[which].p, S.nb, S.NB, S.split, n_bat, (const uint32_t *)counts,
                           (const uint32_t *)block_sums, ctx->wl_list[which].p, S.list_cap, ctx->d_counters + kStatEntries + which, block_sums + wblocks + 40, rank_lo, rank_hi,
                           (uint32_t)width);
        S.list = ctx->wl_list[which].p;
        S.n_list = block_sums + wblocks + 40;   // (NOT the counter block: the compare launch's waves read this word while their atomics hammer that line)
        return FFH_OK;
    };
    FFH_HIP(hipEventRecord(ctx->ev[0], st));
    float ms_cmp = 0, ms_prep = 0;
    unsigned long long cursor_before = 0, n_real_hits = 0;
    // the guides a slab runs on: all of them, then the packed set of those still below the limit
    const uint64_t *act_guides = ctx->guides.p;
    const uint32_t *act_map = nullptr;
    uint32_t n_act = n_guides;
    bool first_launch = true;
    if (bounded) {
        FFH_HIP(ctx->g_total.reserve((size_t)n_guides + 1)); FFH_HIP(ctx->g_flag.reserve((size_t)n_guides + 1)); FFH_HIP(ctx->g_pos.reserve((size_t)n_guides + 2));
        FFH_HIP(ctx->g_active.reserve((size_t)n_guides + 1)); FFH_HIP(ctx->g_map.reserve((size_t)n_guides + 1)); FFH_HIP(ctx->totals.reserve((size_t)n_guides + 1));
        FFH_HIP(hipMemsetAsync(ctx->g_total.p, 0, (size_t)n_guides * 4, st));
    }
    // A bounded scan builds the prefix image's candidate list ONCE, for all guides and all slabs (a slab then is a filter on the
    // work list; a retired guide is made unreachable in the guide table: k_bound_update) instead of enumerating it again, with a
    // counting pass, for every slab.  Needs the whole guide set in one batch and maxMismatch + r1 < prefix width (true of every
    // two-image plan the cost model picks); otherwise the prefix side is binned per slab on the packed active set.
    const bool shared_prefix = bounded && n_guides <= batch && max_mm + plan.r1 < plan.a && !ctx->sw.slab_prefix_per_slab;
    const uint64_t n_items_p_all = (uint64_t)n_guides * (uint64_t)np_p;
    // (Who is retired after a slab is decided on exact position totals: the slab's hits ordered by guide, their target longs
    // gathered, the counts added up, ~0.35 ms per slab.  Round 3 tried a cheaper lower bound -- the compare kernel adding up, per
    // guide, the hits of every (job, group) step that finds two or more: the hits of a repeat family's guides are dense enough to be
    // caught, but positions are what reaches the limit, and a repeat's targets carry counts in the hundreds: 1.3e8 raw hits
    // instead of 4.7e7, 15.4 against 9.6 ms per step on the repeat-structured workload.  Dropped.)
    if (shared_prefix) {
        FFH_HIP(ctx->item_gid.reserve(n_items_p_all + (uint64_t)n_guides * (uint64_t)np_s + 64));
        const int rc = prepare_side(ctx, st, 0, ctx->img[0], plan.r1, ctx->guides.p, -1, n_guides, 0u);
        if (rc) return rc;
    }
    for (size_t sl = 0; sl < slabs.size() && n_act; ++sl) {
        const Slab &SL = slabs[sl];
        const unsigned long long slab_start = cursor_before;
        for (uint32_t g0 = 0; g0 < n_act;) {
            const uint32_t ng = std::min(batch, n_act - g0);
            const uint64_t n_items_p = (uint64_t)ng * (uint64_t)np_p, n_items_s = plan.r2 >= 0 ? (uint64_t)ng * (uint64_t)np_s : 0;
            if (n_items_p + n_items_s >= (1ull << 32) - 64) { ctx->err = "candidate list too large for one batch"; return FFH_E_ARG; }
            FFH_HIP(ctx->item_gid.reserve(n_items_p + n_items_s + 64));
            FFH_HIP(hipEventRecord(ctx->ev[2], st));
            // (The two images' candidate lists do not depend on each other and most of their kernels sit on the launch floor, so round 3
            // ran the suffix image's on a second stream, forked and joined with events: 2.14 against 2.13 ms per step -- the two
            // streams' kernels did not overlap on this stack and every event wait added a few microseconds.  One stream.)
            CompareArgs ca{};
            auto run_prepare = [&]() -> int {
                // the pair counters are per launch (a launch that has to be redone with a larger hit buffer must not count twice)
                hipLaunchKernelGGL(k_compare_setup, dim3(1), dim3(64), 0, st, ctx->d_counters, first_launch ? 1 : 0);
                int rc = FFH_OK;
                if (plan.r2 >= 0) {
                    rc = prepare_side(ctx, st, 1, *SL.suffix, plan.r2, act_guides + g0, -1, ng, (uint32_t)(shared_prefix ? n_items_p_all : n_items_p));
                    if (rc) return rc;
                    rc = side_plan(st, 1, *SL.suffix, SL.n_targets, plan.s, plan.r1, np_s, ng, ca.side[1]);   // a pair with <= r1 mismatches in its prefix is the prefix image's to report
                    if (rc) return rc;
                } else { ca.side[1] = SideArgs{}; ca.side[1].tidx = ctx->img[1].tidx.p; }
                if (!shared_prefix) {
                    rc = prepare_side(ctx, st, 0, ctx->img[0], plan.r1, act_guides + g0, bounded ? -1 : (int64_t)g0, ng, 0u, SL.rank_lo, SL.rank_hi);
                    if (rc) return rc;
                }
                if (shared_prefix) rc = side_plan(st, 0, ctx->img[0], ctx->T, plan.a, -1, np_p, n_guides, ca.side[0], SL.rank_lo, SL.rank_hi, sl == 0);
                else rc = side_plan(st, 0, ctx->img[0], ctx->T, plan.a, -1, np_p, ng, ca.side[0]);
                return rc;
            };
            {
                const bool graphs_on = ctx->sw.graph;
                ffh_ctx::PrepGraph &pg = ctx->pg_slots[ctx->pg_slot];
                const bool eligible = graphs_on && !bounded && !ctx->borrowed && g0 == 0 && ng == n_guides && first_launch;
                // (pattern_gen: the captured kernels read ctx->patterns[side], which prepare_side overwrites IN PLACE -- no reallocation, no
                // epoch change -- when a scan with another (width, radius) comes in between: ADVICE r4.  A plain run that uploads moves
                // the generation on, so only a run that found both lists resident can be followed by a capture, and a capture never
                // contains the upload.)
                const uint64_t key[13] = {(uint64_t)(uintptr_t)(act_guides + g0), ng, (uint64_t)max_mm, (uint64_t)plan.a, (uint64_t)plan.r1, (uint64_t)(int64_t)plan.r2,
                                          (uint64_t)(uintptr_t)ctx->img[0].gwords.p, (uint64_t)(uintptr_t)SL.suffix->gwords.p, ctx->compare_grid, (uint64_t)(uintptr_t)ctx->item_gid.p,
                                          ctx->T, (uint64_t)(uintptr_t)ctx->seg_begin.p, ctx->pattern_gen};
                const uint64_t epoch = prep_signature(ctx, *SL.suffix);
                bool done = false;
                if (eligible && pg.exec && pg.epoch == epoch && !std::memcmp(pg.key, key, sizeof key)) {
                    if (hipGraphLaunch(pg.exec, st) == hipSuccess) {
                        ca.side[0] = pg.side[0]; ca.side[1] = pg.side[1]; expect[0] = pg.expect[0]; expect[1] = pg.expect[1];
                        ctx->n_part[0] = pg.n_part[0]; ctx->n_part[1] = pg.n_part[1];
                        done = true;
                    } else {   // (a replay that cannot be launched: forget the graph, the plain launches below do the work)
                        (void)hipGetLastError();
                        (void)hipGraphExecDestroy(pg.exec);
                        pg.exec = nullptr;
                    }
                } else if (eligible && pg.seen_epoch == epoch && !std::memcmp(pg.seen, key, sizeof key)) {
                    if (pg.exec) { (void)hipGraphExecDestroy(pg.exec); pg.exec = nullptr; }
                    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                        t_capturing = true;
                        const int rc = run_prepare();
                        t_capturing = false;
                        hipGraph_t graph = nullptr;
                        hipError_t e = hipStreamEndCapture(st, &graph);
                        if (rc == FFH_OK && e == hipSuccess && graph) e = hipGraphInstantiate(&pg.exec, graph, nullptr, nullptr, 0);
                        else if (e == hipSuccess) e = hipErrorUnknown;
                        if (graph) (void)hipGraphDestroy(graph);
                        if (e == hipSuccess) e = hipGraphLaunch(pg.exec, st);
                        if (e == hipSuccess) {
                            std::memcpy(pg.key, key, sizeof key); pg.epoch = epoch;
                            pg.side[0] = ca.side[0]; pg.side[1] = ca.side[1]; pg.expect[0] = expect[0]; pg.expect[1] = expect[1];
                            pg.n_part[0] = ctx->n_part[0]; pg.n_part[1] = ctx->n_part[1];
                            done = true;
                        } else {   // (whatever it was: the plain launches below do the work)
                            if (pg.exec) { (void)hipGraphExecDestroy(pg.exec); pg.exec = nullptr; }
                            (void)hipGetLastError();
                            ctx->err.clear();
                        }
                    } else (void)hipGetLastError();
                }
                if (!done) {
                    const int rc = run_prepare();
                    if (rc) return rc;
                    std::memcpy(pg.seen, key, sizeof key);
                    pg.seen_epoch = prep_signature(ctx, *SL.suffix);
                }
            }
            FFH_HIP(hipEventRecord(ctx->ev[3], st));   // (prepare_ms: candidate lists and work lists; compare_ms: the compare launch alone)
            ca.gids = ctx->item_gid.p; ca.hits = ctx->hits.p; ca.cap = (uint64_t)ctx->hits.cap; ca.tbits = ctx->tbits; ca.max_mm = max_mm;
            ca.guide_base[0] = shared_prefix ? 0u : g0; ca.guide_base[1] = g0;
            ca.gmap[0] = shared_prefix ? nullptr : (act_map ? act_map + g0 : nullptr);
            ca.gmap[1] = act_map ? act_map + g0 : nullptr;
            // how the launch deals its work entries: queue chunks of 16 for long lists; chunks of 4 for the medium-length lists of a bounded
            // scan's slabs, whose entries differ widely in weight (repeat families: 4.11 against 4.30 ms of compare per step); a fixed
            // stride otherwise -- the entries of a uniform medium list (a bin shard: an eighth of hg38) weigh the same, and there the
            // queue's draws only cost (0.282 against 0.238 ms per launch)
            int chunk = plan.r2 < 0 ? work_list_chunk(expect[0], ctx->compare_grid) : std::min(work_list_chunk(expect[0], ctx->compare_grid), work_list_chunk(expect[1], ctx->compare_grid));
            if (!bounded && chunk < (int)kQueueChunkLong) chunk = 0;
            if (!launch_compare(ca, ctx->d_counters, ctx->compare_grid, st, chunk, ctx->sw.generic_compare, ctx->sw.work_queue)) {
                ctx->err = "no compare kernel for rest keys of " + std::to_string(ca.side[0].rest) + " + " + std::to_string(ca.side[1].rest) + " bases";
                return FFH_E_STATE;
            }
            FFH_HIP(hipGetLastError());
            FFH_HIP(hipEventRecord(ctx->ev[4], st));
            unsigned long long cnt[16];  // one read-back: hit cursor, hit count, executed pairs and work entries of the two images
            FFH_HIP(spin_wait(ctx, cnt));
            FFH_HIP(hipGetLastError());
            first_launch = false;
            if (FFH_TRIP_STATS) {   // (variant builds only: tools/build_variant.sh ... FFH_TRIP_STATS=1)
                unsigned long long ts[14];
                FFH_HIP(hipMemcpy(ts, ctx->d_counters + 16, sizeof ts, hipMemcpyDeviceToHost));
                fprintf(stderr, "[trip stats] suffix: rows %llu steps %llu parks %llu pushes %llu hit_steps %llu lane_steps %llu | prefix: rows %llu steps %llu parks %llu pushes %llu hit_steps %llu "
                        "lane_steps %llu | flushes %llu flush_iterations %llu | entries %llu %llu hits %llu\n", ts[6], ts[7], ts[8], ts[9], ts[10], ts[11], ts[0], ts[1], ts[2], ts[3], ts[4], ts[5],
                        ts[12], ts[13], cnt[kStatEntries], cnt[kStatEntries + 1], cnt[1]);
            }
            const unsigned long long cursor = cnt[0];
            // segments, sort offsets and the epilogue index hits with 32 bits: ONE scan never holds more raw hits than that.
            // ffh_discover / ffh_discover_sharded / ffh_discover_bulge then bound the scan and, if that is not enough, split the guide set
            // and merge the parts' results (discover_split): the reference is slow on such a guide set, not wrong
            // (BlockManager.scala:212-254), so the library does not refuse it either.  A caller of the two-step ffh_scan / ffh_finalize
            // gets this error and splits itself.
            if (cursor >= ctx->sw.raw_hit_limit) {
                ctx->too_many_hits = true;
                ctx->err = "more than 2^32 raw hits in one scan: ffh_discover splits such a guide set by itself; with ffh_scan / ffh_finalize pass fewer guides per call";
                return FFH_E_ARG;
            }
            bool redo = false;
            if (cursor > ctx->hits.cap) {  // hit buffer too small: grow it and redo this batch (earlier batches are kept, copied device to device)
                DevBuf<uint64_t> bigger;
                FFH_HIP(bigger.reserve((size_t)(cursor + cursor / 2)));
                if (cursor_before) FFH_HIP(hipMemcpyAsync(bigger.p, ctx->hits.p, (size_t)cursor_before * 8, hipMemcpyDeviceToDevice, st));
                FFH_HIP(hipStreamSynchronize(st));
                ctx->hits = std::move(bigger);
                redo = true;
            }
            for (int w = 0; w < 2; ++w)   // a work list that did not hold all entries (candidates piled on a few buckets): the same
                if (cnt[kStatEntries + w] > ctx->wl_list[w].cap) {
                    FFH_HIP(hipStreamSynchronize(st));
                    FFH_HIP(ctx->wl_list[w].reserve((size_t)(cnt[kStatEntries + w] + cnt[kStatEntries + w] / 4 + 64)));
                    redo = true;
                }
            if (redo) {
                const unsigned long long back[2] = {cursor_before, n_real_hits};
                FFH_HIP(hipMemcpy(ctx->d_counters, back, 16, hipMemcpyHostToDevice));  // the hit cursor and the hit count go back to where this batch began
                continue;
            }
            ctx->tm.pairs_prefix += cnt[kStatPairs]; ctx->tm.pairs_suffix += cnt[kStatPairs + 1];
            float a = 0, b = 0;
            FFH_HIP(hipEventSynchronize(ctx->ev[4]));   // (complete on the device -- the counters behind it have arrived -- but the event itself may not be marked yet)
            FFH_HIP(hipEventElapsedTime(&a, ctx->ev[2], ctx->ev[3]));
            FFH_HIP(hipEventElapsedTime(&b, ctx->ev[3], ctx->ev[4]));
            ms_prep += a; ms_cmp += b;
            ctx->tm.items_prefix += (uint64_t)((double)ng * np_p); ctx->tm.tiles_prefix += cnt[kStatEntries];
            ctx->tm.items_suffix += (uint64_t)((double)ng * np_s); ctx->tm.tiles_suffix += cnt[kStatEntries + 1];
            ctx->tm.compare_launches++;
            cursor_before = cursor;  // the records already are sort keys: (global guide << tbits) | database index
            n_real_hits = cnt[1];    // the waves' own count (the cursor includes the padding of their last chunks)
            g0 += ng;
        }
        if (sl + 1 == slabs.size()) break;
        // ---- the slab's positions per guide -> who is still below the limit -> the packed guide set of the next slab ----
        // (Round 5 tried ordering every slab completely as it ends -- k_segsort + k_segsort_heavy per slab, the target longs kept -- and
        // concatenating the slabs' segments per guide instead of the final device-wide sort: 10.0 against 8.3 ms per step of the
        // repeat-structured workload.  A wave per guide has a floor of ~0.22 ms per launch, paid six times, and the guides inside
        // repeat families went through the block-level sort six times: profiles/r05/ab_log.txt 4.)
        const uint64_t n_new = cursor_before - slab_start;
        FFH_HIP(hipMemsetAsync(ctx->seg_begin.p, 0, (size_t)n_guides * 4, st));
        FFH_HIP(hipMemsetAsync(ctx->seg_end.p, 0, (size_t)n_guides * 4, st));
        if (n_new) {
            uint64_t *sp = ctx->hits.p + slab_start;
            if (n_new <= kSmallSort) hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(1024), 0, st, sp, (uint32_t)n_new);
            else {
                const uint32_t nbk = sort_nblocks(n_new);
                FFH_HIP(ctx->hits_alt.reserve(ctx->hits.cap));
                FFH_HIP(ctx->sort_table.reserve((size_t)kSortTableDigits * nbk + 1));
                FFH_HIP(ctx->sort_offs.reserve((size_t)kSortTableDigits * nbk + 1));
                FFH_HIP(ctx->scan_tmp32.reserve(scan_scratch_elems_safe((uint64_t)kSortTableDigits * nbk)));
                SortScratch ss;
                ss.alt = ctx->hits_alt.p + slab_start; ss.table = ctx->sort_table.p; ss.offs = ctx->sort_offs.p; ss.scan_tmp = ctx->scan_tmp32.p;
                // by guide only (two passes instead of five): the totals do not depend on the order inside a guide's segment
                sp = radix_sort_u64(sp, n_new, ctx->tbits, ctx->tbits + gbits, 64, 64, ss, st);   // (either buffer then holds a permutation of the slab's records)
            }
            hipLaunchKernelGGL(k_segments, dim3(blocks_for(n_new, 256)), dim3(256), 0, st, (const uint64_t *)sp, n_new, ctx->tbits, n_guides, ctx->seg_begin.p, ctx->seg_end.p);
            FFH_HIP(ctx->hit_t.reserve(n_new + 1));
            hipLaunchKernelGGL(k_hit_targets, dim3(blocks_for(n_new, 256)), dim3(256), 0, st, (const uint64_t *)sp, n_new, ctx->tbits, n_guides, (const uint64_t *)ctx->targets.p, ctx->hit_t.p);
        }
        hipLaunchKernelGGL(k_cutoff, dim3(blocks_for(n_guides, 4)), dim3(256), 0, st, ctx->seg_begin.p, ctx->seg_end.p, (const uint64_t *)ctx->hit_t.p, (const uint32_t *)nullptr,
                           n_guides, bound_ot, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, ctx->totals.p, (uint32_t *)nullptr);
        hipLaunchKernelGGL(k_bound_update, dim3(blocks_for(n_guides, 256)), dim3(256), 0, st, ctx->g_total.p, (const uint32_t *)ctx->totals.p, n_guides, bound_ot, ctx->g_flag.p,
                           shared_prefix ? ctx->gtab[0].p : (uint2 *)nullptr, plan.a >= 16 ? 0xFFFFFFFFu : (1u << (2 * plan.a)) - 1u);
        FFH_HIP(ctx->scan_tmp32.reserve(scan_scratch_elems_safe(n_guides)));
        exclusive_scan<uint32_t, uint32_t>(ctx->g_flag.p, n_guides, ctx->g_pos.p, ctx->scan_tmp32.p, st);
        hipLaunchKernelGGL(k_bound_compact, dim3(blocks_for(n_guides, 256)), dim3(256), 0, st, (const uint64_t *)ctx->guides.p, (const uint32_t *)ctx->g_flag.p,
                           (const uint32_t *)ctx->g_pos.p, n_guides, ctx->g_active.p, ctx->g_map.p);
        FFH_HIP(hipGetLastError());
        uint32_t still = 0;
        FFH_HIP(spin_wait(ctx, nullptr, ctx->g_pos.p + n_guides, &still));
        ctx->tm.retired_guides = n_guides - still;
        act_guides = ctx->g_active.p; act_map = ctx->g_map.p; n_act = still;
    }
    ctx->n_raw = cursor_before;
    FFH_HIP(hipEventRecord(ctx->ev[5], st));
    if (bounded) {   // (k_guide_keys cleared nothing: the slabs' own segments are in the arrays)
        FFH_HIP(hipMemsetAsync(ctx->seg_begin.p, 0, (size_t)n_guides * 4, st));
        FFH_HIP(hipMemsetAsync(ctx->seg_end.p, 0, (size_t)n_guides * 4, st));
    }
    // ---- order the hits by (guide, database index) ----
    {
        const int rc = order_hits(ctx, st, ctx->hits.p, ctx->hits_alt, 0, ctx->n_raw, n_real_hits, gbits, n_guides, ctx->seg_begin.p, ctx->seg_end.p, &ctx->hits_sorted, &ctx->n_raw);
        if (rc) return rc;
    }
    ctx->hit_t_ready = false;  // the target longs of the hits are gathered on demand (gather_hit_targets)
    FFH_HIP(hipEventRecord(ctx->ev[6], st));
    FFH_HIP(hipGetLastError());
    // no synchronisation here: the ordering kernels run while the caller comes back with ffh_finalize / ffh_shard_totals (same
    // stream); their timings are read when somebody asks for them (finish_scan_timings)
    ctx->tm.prepare_ms = ms_prep; ctx->tm.compare_ms = ms_cmp; ctx->tm.n_raw_hits = n_real_hits;
    ctx->scan_timing_pending = true;
    ctx->scanned = true;
    // a guide set that sits in repeat families (thousands of raw hits per guide, most of them beyond any cut-off): bound the
    // scans that follow on this context
    constexpr unsigned long long kBoundAutoHits = 2048;
    if (ctx->bound_auto && !ctx->bound_mode && n_guides >= 64 && n_real_hits > kBoundAutoHits * n_guides && ctx->slabs_state >= 0) ctx->bound_mode = 1;
    return FFH_OK;
}

int ffh_scan(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mm) { return scan_impl(ctx, guides, n_guides, max_mm, 0u); }

int ffh_scan_bounded(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mm, int max_offtargets) {
    if (max_offtargets < 0) { if (ctx) ctx->err = "bad argument"; return FFH_E_ARG; }
    return scan_impl(ctx, guides, n_guides, max_mm, (ctx && ctx->bound_mode) ? (uint32_t)max_offtargets : 0u);
}

int ffh_set_bounding(ffh_ctx *ctx, int mode) {
    if (!ctx || mode < -1 || mode > 1) return FFH_E_ARG;
    ctx->bound_auto = mode < 0;
    ctx->bound_mode = mode > 0 ? 1 : 0;
    return FFH_OK;
}

// a bounded scan holds, for a retired guide, only the hits up to the slab in which it reached bound_ot positions
// -- which is everything a caller with a limit <= bound_ot can ask for.  A larger limit (ffh_discover(A) followed by ffh_finalize /
// ffh_shard_totals with B > A) needs hits the bounded scan never collected: the guide set is still resident, so the scan is redone
// unbounded instead of making the answer depend on whether an earlier call happened to switch bounding on.
static int check_bound(ffh_ctx *ctx, int64_t limit) {
    if (!ctx->bound_ot || limit <= (int64_t)ctx->bound_ot) return FFH_OK;
    return scan_impl(ctx, ctx->guides.p, ctx->n_guides, ctx->max_mm, 0u);
}

// hit_t[i] = target long of sorted hit i: needed by the paths that deliver hit lists or shard totals; the aggregates-only epilogue
// gathers on the fly
static int gather_hit_targets(ffh_ctx *ctx) {
    if (ctx->hit_t_ready) return FFH_OK;
    FFH_HIP(ctx->hit_t.reserve(ctx->n_raw + 1));
    if (ctx->n_raw)
        hipLaunchKernelGGL(k_hit_targets, dim3(blocks_for(ctx->n_raw, 256)), dim3(256), 0, ctx->st, ctx->hits_sorted, ctx->n_raw, ctx->tbits, ctx->n_guides, ctx->targets.p, ctx->hit_t.p);
    FFH_HIP(hipGetLastError());
    ctx->hit_t_ready = true;
    return FFH_OK;
}

static void finish_scan_timings(ffh_ctx *ctx) {
    if (!ctx->scan_timing_pending) return;
    ctx->scan_timing_pending = false;
    (void)hipSetDevice(ctx->device);
    if (hipEventSynchronize(ctx->ev[6]) != hipSuccess) return;
    float ms_sort = 0, ms_total = 0;
    (void)hipEventElapsedTime(&ms_sort, ctx->ev[5], ctx->ev[6]);
    (void)hipEventElapsedTime(&ms_total, ctx->ev[0], ctx->ev[6]);
    ctx->tm.sort_ms = ms_sort; ctx->tm.total_scan_ms = ms_total;
}
static void finish_finalize_timing(ffh_ctx *ctx) {  // the stream-ordered shard epilogue leaves its two events behind
    if (!ctx->finalize_timing_pending) return;
    ctx->finalize_timing_pending = false;
    (void)hipSetDevice(ctx->device);
    if (hipEventSynchronize(ctx->ev[1]) != hipSuccess) return;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev[7], ctx->ev[1]);
    ctx->tm.finalize_ms = ms;
}

int ffh_shard_totals(ffh_ctx *ctx, uint32_t *totals, uint32_t clamp) {
    if (!ctx || !totals) return FFH_E_ARG;
    if (!ctx->scanned) { ctx->err = "ffh_scan has not run"; return FFH_E_STATE; }
    { const int rc = check_bound(ctx, clamp); if (rc) return rc; }
    FFH_HIP(hipSetDevice(ctx->device));
    FFH_HIP(ctx->totals.reserve((size_t)ctx->n_guides + 1));
    { const int rc = gather_hit_targets(ctx); if (rc) return rc; }
    if (ctx->n_guides) {
        hipLaunchKernelGGL(k_cutoff, dim3(blocks_for(ctx->n_guides, 4)), dim3(256), 0, ctx->st, ctx->seg_begin.p, ctx->seg_end.p, ctx->hit_t.p, (const uint32_t *)nullptr,
                           ctx->n_guides, clamp, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, ctx->totals.p, (uint32_t *)nullptr);
        FFH_HIP(hipMemcpyAsync(totals, ctx->totals.p, (size_t)ctx->n_guides * 4, hipMemcpyDeviceToHost, ctx->st));
    }
    FFH_HIP(hipStreamSynchronize(ctx->st));
    return FFH_OK;
}

int ffh_shard_totals_device(ffh_ctx *ctx, uint32_t *device_totals, uint32_t clamp) {
    if (!ctx || !device_totals) return FFH_E_ARG;
    if (!ctx->scanned) { ctx->err = "ffh_scan has not run"; return FFH_E_STATE; }
    { const int rc = check_bound(ctx, clamp); if (rc) return rc; }
    FFH_HIP(hipSetDevice(ctx->device));
    FFH_HIP(hipDeviceSynchronize());  // the caller's buffer may still be written by another stream (its allocation's fill, a collective)
    { const int rc = gather_hit_targets(ctx); if (rc) return rc; }
    if (ctx->n_guides)
        hipLaunchKernelGGL(k_cutoff, dim3(blocks_for(ctx->n_guides, 4)), dim3(256), 0, ctx->st, ctx->seg_begin.p, ctx->seg_end.p, ctx->hit_t.p, (const uint32_t *)nullptr,
                           ctx->n_guides, clamp, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, device_totals, (uint32_t *)nullptr);
    FFH_HIP(hipGetLastError());
    FFH_HIP(hipStreamSynchronize(ctx->st));
    return FFH_OK;
}

int ffh_summaries_to_device(ffh_ctx *ctx, void *device_summaries) {
    if (!ctx || !device_summaries) return FFH_E_ARG;
    if (!ctx->scanned) { ctx->err = "no finalized scan"; return FFH_E_STATE; }
    FFH_HIP(hipSetDevice(ctx->device));
    FFH_HIP(hipDeviceSynchronize());
    if (ctx->n_guides) FFH_HIP(hipMemcpyAsync(device_summaries, ctx->summ.p, (size_t)ctx->n_guides * sizeof(ffh_guide_summary), hipMemcpyDeviceToDevice, ctx->st));
    FFH_HIP(hipStreamSynchronize(ctx->st));
    return FFH_OK;
}

// device -> page-locked host.  (A copy kernel of our own with a few persistent blocks storing into the mapped result block was
// tried in place of the runtime's copy, which is a kernel too (__amd_rocclr_copyBuffer): 16 blocks reach 36 GB/s and leave the
// kernels beside them alone, 64 blocks reach the runtime's 53-55 GB/s and slow them 5-30x exactly as the runtime's copy does --
// it is the host-bound write traffic, not the CUs it occupies.  End to end the runtime's copy was 0.4-0.7 ms faster.)
static hipError_t copy_out(ffh_ctx *, void *host, const void *dev, size_t bytes, hipStream_t st) {
    return bytes ? hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st) : hipSuccess;
}

// ---- the list-delivering half of ffh_finalize ------------------------------------------------------------------------------------
// One call delivers the lists of the guides of the CURRENT scan into `lp.r`.  Alone (ffh_finalize): the result is allocated here, exactly,
// and the call returns when everything has arrived.  As a part of a pipelined ffh_discover (discover_pipelined, round 5): the guide set
// is scanned in two halves; the first half's per-hit arrays are still crossing the link (~55 GB/s, ~1 ms per half at hg38 scale) while
// the second half is scanned -- a device-to-host copy next to the latency-bound compare launch slows neither
// (profiles/r05/overlap_probe.json: 1.82 ms step + 1.84 ms copy = 2.05 ms together).  The halves write one result: per-guide arrays at
// g_base, per-hit arrays at h_base, positions at p_base, on the device and on the host; the first half sizes the block for both from
// its own counts, and a second half that does not fit abandons the pipeline (the caller then runs the unsplit call).
struct ListPipe {
    ffh_result *r = nullptr;
    uint32_t G_total = 0, g_base = 0;
    uint64_t h_base = 0, p_base = 0, h_cap = 0, p_cap = 0;
    bool pipelined = false, last = true;
    uint64_t Hr = 0, Pr = 0;   // this part's counts (out)
    bool overflow = false;     // out: the second half did not fit the block the first one sized
};
static int finalize_lists(ffh_ctx *ctx, const uint32_t *d_prior, int max_offtargets, unsigned flags, ListPipe &lp) {
    hipStream_t st = ctx->st;
    const uint32_t G = ctx->n_guides;
    if (lp.pipelined) {   // (ffh_finalize did this already for a call of its own)
        FFH_HIP(hipEventRecord(ctx->ev[7], st));
        FFH_HIP(ctx->n_ret.reserve((size_t)G + 1)); FFH_HIP(ctx->ot_count.reserve((size_t)G + 1)); FFH_HIP(ctx->full.reserve((size_t)G + 1));
        FFH_HIP(ctx->ret_off.reserve((size_t)G + 2)); FFH_HIP(ctx->summ.reserve((size_t)G + 1));
        FFH_HIP(ctx->scan_tmp64.reserve(scan_scratch_elems_safe(std::max<uint64_t>(G, ctx->n_raw) + 1)));
    }
    { const int rc = gather_hit_targets(ctx); if (rc) return rc; }
    const bool want_pos = !(flags & FFH_FINALIZE_NO_POSITIONS), want_cfd = !(flags & FFH_FINALIZE_NO_HIT_SCORES);
    // ordered cut-off; with positions wanted it also leaves, per kept hit, the number of the guide's kept positions before it, so that
    // every hit's slot in the position array follows from one scan over the guides (no scan over the hits, no second round trip)
    if (want_pos) { FFH_HIP(ctx->hit_pre.reserve(ctx->n_raw + 1)); FFH_HIP(ctx->pos_base.reserve((size_t)G + 2)); }
    if (G) hipLaunchKernelGGL(k_cutoff, dim3(blocks_for(G, 4)), dim3(256), 0, st, ctx->seg_begin.p, ctx->seg_end.p, ctx->hit_t.p, d_prior, G, (uint32_t)max_offtargets,
                              ctx->n_ret.p, ctx->ot_count.p, ctx->full.p, (uint32_t *)nullptr, want_pos ? ctx->hit_pre.p : (uint32_t *)nullptr);
    exclusive_scan<uint32_t, uint64_t>(ctx->n_ret.p, G, ctx->ret_off.p, ctx->scan_tmp64.p, st);
    uint64_t Hr = 0, Pr = 0;
    FFH_HIP(hipMemcpyAsync(&Hr, ctx->ret_off.p + G, 8, hipMemcpyDeviceToHost, st));
    if (want_pos) {
        exclusive_scan<uint32_t, uint64_t>(ctx->ot_count.p, G, ctx->pos_base.p, ctx->scan_tmp64.p, st);
        FFH_HIP(hipMemcpyAsync(&Pr, ctx->pos_base.p + G, 8, hipMemcpyDeviceToHost, st));
    }
    FFH_HIP(hipStreamSynchronize(st));
    lp.Hr = Hr; lp.Pr = Pr;
    const uint64_t hb = lp.h_base, pb = lp.p_base;
    if (!lp.r) {   // the only part, or the first: the result block (and the device arrays) for everything
        uint64_t hc = Hr, pc = Pr;
        if (lp.pipelined) {   // room for the other half: this half's counts scaled to the whole guide set, + 30 %
            const double f = 1.3 * (double)lp.G_total / (double)std::max<uint32_t>(G, 1u);
            hc = (uint64_t)((double)Hr * f) + 4096; pc = (uint64_t)((double)Pr * f) + 4096;
        }
        lp.h_cap = hc; lp.p_cap = pc;
        ffh_result *r = new (std::nothrow) ffh_result();
        if (!r || !r->allocate(ctx->pool, lp.G_total, hc, true, want_cfd, want_pos) || (want_pos && !r->allocate_positions(pc))) {
            delete r; ctx->err = "out of (pinned) host memory"; return FFH_E_NOMEM;
        }
        r->scores_valid = ctx->geo.cas9_23;
        r->pos_offsets_pending = want_pos;  // never copied: hit h owns (hit_targets[h] >> 48) positions (settle_pos_offsets)
        lp.r = r;
    } else if (hb + Hr > lp.h_cap || pb + Pr > lp.p_cap) { lp.overflow = true; return FFH_OK; }
    ffh_result *r = lp.r;
    FFH_HIP(ctx->out_target.reserve(lp.h_cap + 2));
    FFH_HIP(ctx->out_mm.reserve(lp.h_cap + 16));
    FFH_HIP(ctx->out_cnt.reserve(std::max<uint64_t>(lp.h_cap, ctx->T) + 1));
    FFH_HIP(ctx->out_tidx.reserve(lp.h_cap + 1));
    FFH_HIP(ctx->out_cfd.reserve(lp.h_cap + 2));
    FFH_HIP(ctx->out_hsu.reserve(lp.h_cap + 1));
    double *d_jost = nullptr;  // the CRISPRi aggregates are computed on request only: they cost a third per-hit array
    if (flags & FFH_FINALIZE_JOST) { FFH_HIP(ctx->out_jost.reserve(lp.h_cap + 1)); d_jost = ctx->out_jost.p + hb; }
    if (want_pos) { FFH_HIP(ctx->out_posoff.reserve(lp.h_cap + 2)); FFH_HIP(ctx->out_pos.reserve(lp.p_cap + 2)); }
    if (ctx->n_raw)
        hipLaunchKernelGGL(k_score_hits, dim3(blocks_for(ctx->n_raw, 256)), dim3(256), 0, st, ctx->hits_sorted, ctx->n_raw, ctx->tbits, G, ctx->seg_begin.p, ctx->n_ret.p, ctx->ret_off.p,
                           ctx->hit_t.p, ctx->guides.p, ctx->geo, ctx->d_tab, ctx->out_target.p + hb, ctx->out_mm.p + hb, ctx->out_cnt.p + hb, ctx->out_tidx.p + hb, ctx->out_cfd.p + hb,
                           ctx->out_hsu.p + hb, d_jost, want_pos ? (const uint32_t *)ctx->hit_pre.p : (const uint32_t *)nullptr,
                           want_pos ? (const uint64_t *)ctx->pos_base.p : (const uint64_t *)nullptr, want_pos ? ctx->out_posoff.p + hb : (uint64_t *)nullptr);
    auto fail = [&](const char *what, hipError_t e) {
        (void)hipStreamSynchronize(ctx->copy_st); (void)hipStreamSynchronize(st);
        ctx->err = std::string(what) + hipGetErrorString(e); delete lp.r; lp.r = nullptr;
        return FFH_E_HIP;
    };
    auto after_main = [&]() {   // the copy stream goes on when the main stream has come this far
        hipError_t e = hipEventRecord(ctx->copy_ev, st);
        return e == hipSuccess ? hipStreamWaitEvent(ctx->copy_st, ctx->copy_ev, 0) : e;
    };
    auto copy_hits = [&]() {
        hipError_t e = hipSuccess;
        if (Hr) e = copy_out(ctx, r->hit_targets + hb, ctx->out_target.p + hb, Hr * 8, ctx->copy_st);
        if (Hr && e == hipSuccess) e = copy_out(ctx, r->hit_mm + hb, ctx->out_mm.p + hb, Hr, ctx->copy_st);
        if (Hr && want_cfd && e == hipSuccess) e = copy_out(ctx, r->hit_cfd + hb, ctx->out_cfd.p + hb, Hr * 8, ctx->copy_st);
        return e;
    };
    auto gather_positions = [&]() {
        if (want_pos && Hr) hipLaunchKernelGGL(k_gather_positions, dim3(blocks_for(Hr, 256)), dim3(256), 0, st, ctx->out_tidx.p + hb, ctx->out_cnt.p + hb, ctx->out_posoff.p + hb, Hr,
                                               ctx->pos_off.p, ctx->positions.p, ctx->out_pos.p + pb);
    };
    auto aggregate = [&]() {
        if (G) hipLaunchKernelGGL(k_guide_aggregate, dim3(blocks_for(G, 4)), dim3(256), 0, st, ctx->ret_off.p, ctx->n_ret.p, ctx->ot_count.p, ctx->full.p, ctx->out_mm.p + hb,
                                  ctx->out_cnt.p + hb, ctx->out_cfd.p + hb, ctx->out_hsu.p + hb, (const double *)d_jost, G, ctx->summ.p);
    };
    hipError_t e = hipSuccess;
    if (lp.last) {
        // The link (~55 GB/s) is what a list-delivering call waits for: the per-hit arrays leave on the copy stream as soon as
        // k_score_hits has written them; the positions are gathered beside that transfer and queue behind it; the aggregation and the
        // small per-guide copies run on the main stream meanwhile (both kernels run 5-10 x slower beside the runtime's copy kernel
        // than alone, and are still done before it is).
        e = after_main();
        if (e == hipSuccess) e = copy_hits();
        if (e != hipSuccess) return fail("result copy: ", e);
        gather_positions();
        if (want_pos) {
            e = after_main();
            if (Pr && e == hipSuccess) e = copy_out(ctx, r->positions + pb, ctx->out_pos.p + pb, Pr * 8, ctx->copy_st);
        }
        aggregate();
    } else {
        // a half that another half follows: its kernels first, ALONE (behind them waits the next half's scan, and beside the copy kernel
        // they would take 1.0-1.3 ms each instead of 0.1-0.25), then all its copies, which the next half's scan runs beside at no cost
        gather_positions();
        aggregate();
        // (the per-guide arrays leave on the copy stream too, from device copies made HERE: the next half's kernels overwrite summ /
        // ret_off, a copy of them on the main stream would queue for the link in front of the next half's scan, and even a
        // device-to-device copy started beside the transfers below takes 0.65 ms instead of a few microseconds)
        FFH_HIP(ctx->summ_stage.reserve((size_t)G + 1)); FFH_HIP(ctx->ret_off_stage.reserve((size_t)G + 2));
        if (G) e = hipMemcpyAsync(ctx->summ_stage.p, ctx->summ.p, (size_t)G * sizeof(ffh_guide_summary), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->ret_off_stage.p, ctx->ret_off.p, ((size_t)G + 1) * 8, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev[1], st);
        if (e == hipSuccess) e = after_main();
        if (e == hipSuccess) e = copy_hits();
        if (want_pos && Pr && e == hipSuccess) e = copy_out(ctx, r->positions + pb, ctx->out_pos.p + pb, Pr * 8, ctx->copy_st);
        if (G && e == hipSuccess) e = hipMemcpyAsync(r->summaries + lp.g_base, ctx->summ_stage.p, (size_t)G * sizeof(ffh_guide_summary), hipMemcpyDeviceToHost, ctx->copy_st);
        // (G entries, not G + 1: the next part writes entry g_base + G itself, and this copy may land after that one)
        if (G && e == hipSuccess) e = hipMemcpyAsync(r->guide_offsets + lp.g_base, ctx->ret_off_stage.p, (size_t)G * 8, hipMemcpyDeviceToHost, ctx->copy_st);
    }
    if (e != hipSuccess) return fail("result copy: ", e);
    if (lp.last) {
        if (e == hipSuccess) e = hipEventRecord(ctx->ev[1], st);
        if (G && e == hipSuccess) e = hipMemcpyAsync(r->summaries + lp.g_base, ctx->summ.p, (size_t)G * sizeof(ffh_guide_summary), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(r->guide_offsets + lp.g_base, ctx->ret_off.p, ((size_t)G + 1) * 8, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail("result copy: ", e);
    if (!lp.last) return FFH_OK;   // (the next half's scan goes on behind these launches; its finalize_lists waits for everything)
    e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_st);
    if (e != hipSuccess) return fail("result copy: ", e);
    if (hb)   // the second half's offsets count from its own first hit
        for (uint32_t g = 0; g <= G; ++g) r->guide_offsets[lp.g_base + g] += hb;
    r->n_hits = hb + Hr;
    r->n_positions = want_pos ? pb + Pr : 0;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev[7], ctx->ev[1]);
    ctx->tm.finalize_ms = ms;
    finish_scan_timings(ctx);
    return FFH_OK;
}

int ffh_finalize(ffh_ctx *ctx, const uint32_t *prior_totals, int max_offtargets, unsigned flags, ffh_result **out) {
    if (!ctx || !out || max_offtargets < 0) { if (ctx) ctx->err = "bad argument"; return FFH_E_ARG; }
    if (!ctx->scanned) { ctx->err = "ffh_scan has not run"; return FFH_E_STATE; }
    { const int rc = check_bound(ctx, max_offtargets); if (rc) return rc; }
    FFH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->st;
    const uint32_t G = ctx->n_guides;
    FFH_HIP(hipEventRecord(ctx->ev[7], st));
    FFH_HIP(ctx->n_ret.reserve((size_t)G + 1));
    FFH_HIP(ctx->ot_count.reserve((size_t)G + 1));
    FFH_HIP(ctx->full.reserve((size_t)G + 1));
    FFH_HIP(ctx->ret_off.reserve((size_t)G + 2));
    FFH_HIP(ctx->summ.reserve((size_t)G + 1));
    FFH_HIP(ctx->scan_tmp64.reserve(scan_scratch_elems_safe(std::max<uint64_t>(G, ctx->n_raw) + 1)));
    const uint32_t *d_prior = nullptr;
    if (prior_totals) {
        FFH_HIP(ctx->prior.reserve((size_t)G + 1));
        if (flags & FFH_FINALIZE_PRIOR_ON_DEVICE) FFH_HIP(hipDeviceSynchronize());  // the producer (an RCCL collective, a torch op) used another stream
        if (G) FFH_HIP(hipMemcpyAsync(ctx->prior.p, prior_totals, (size_t)G * 4, (flags & FFH_FINALIZE_PRIOR_ON_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        d_prior = ctx->prior.p;
    }
    static_assert(sizeof(GuideSummary) == sizeof(ffh_guide_summary), "summary layouts must agree");
    if (flags & FFH_FINALIZE_SUMMARIES_ONLY) {
        // aggregates only: one fused pass per guide (cut-off, scores, ordered sums), no per-hit arrays, one synchronisation
        ffh_result *r = new (std::nothrow) ffh_result();
        if (!r || !r->allocate(ctx->pool, G, 0, false)) { delete r; ctx->err = "out of (pinned) host memory"; return FFH_E_NOMEM; }
        r->scores_valid = ctx->geo.cas9_23;
        // the kernel stores every summary into the result's page-locked block as well (hipHostMalloc memory is mapped into the
        // device's address space): the 88 bytes per guide cross the link under the kernel instead of in a copy after it
        const bool zero_copy = !ctx->sw.summary_copy;
        if (G) hipLaunchKernelGGL(k_guide_epilogue, dim3(blocks_for(G, 4)), dim3(256), 0, st, ctx->seg_begin.p, ctx->seg_end.p,
                                  (const uint64_t *)(ctx->hit_t_ready ? ctx->hit_t.p : nullptr), (const uint64_t *)ctx->hits_sorted, (const uint64_t *)ctx->targets.p, ctx->tbits,
                                  d_prior, ctx->guides.p, ctx->geo,
                                  ctx->d_tab, G, (uint32_t)max_offtargets, (flags & FFH_FINALIZE_JOST) ? 1 : 0, ctx->n_ret.p, ctx->summ.p, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                  zero_copy ? (GuideSummary *)r->summaries : (GuideSummary *)nullptr);
        // no scan of the per-guide hit counts and no copy of the offsets: nobody needs them to read the aggregates, and whoever
        // asks (ffh_result_guide_offsets / ffh_result_n_hits) gets them folded from the summaries' n_hits on the host
        hipError_t e = hipEventRecord(ctx->ev[1], st);
        if (e == hipSuccess) e = hipGetLastError();
        if (G && e == hipSuccess && !zero_copy) e = hipMemcpyAsync(r->summaries, ctx->summ.p, (size_t)G * sizeof(ffh_guide_summary), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = spin_wait(ctx, nullptr);
        if (e != hipSuccess) { ctx->err = std::string("finalize: ") + hipGetErrorString(e); delete r; return FFH_E_HIP; }
        r->offsets_pending = true;
        float ms = 0;
        (void)hipEventSynchronize(ctx->ev[1]);
        (void)hipEventElapsedTime(&ms, ctx->ev[7], ctx->ev[1]);
        ctx->tm.finalize_ms = ms;
        finish_scan_timings(ctx);
        *out = r;
        return FFH_OK;
    }
    ListPipe lp;
    lp.G_total = G;
    const int rc = finalize_lists(ctx, d_prior, max_offtargets, flags, lp);
    if (rc) return rc;
    *out = lp.r;
    return FFH_OK;
}

// ---- a scan the reference would finish is never refused -------------------------------------------------------------------------------
// One scan holds fewer than 2^32 raw hits (32-bit segment arithmetic).  A guide set that collects more -- <= 5 or 6 mismatches on a
// repeat-rich genome -- is first scanned BOUNDED (what the auto rule switches on after any scan with more than 2048 raw hits per guide:
// guides that have reached maximumOffTargets are retired slab by slab, as the reference stops feeding a full guide,
// crispr/ResultsAggregator.scala:61-69), and if that still overflows the guide set is halved, the halves are discovered one after the
// other and their results concatenated (guides are independent of each other everywhere on the path).
static int scan_retry_bounded(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets) {
    int rc = ffh_scan_bounded(ctx, guides, n_guides, max_mismatch, max_offtargets);
    if (rc && ctx && ctx->too_many_hits && !ctx->bound_mode && ctx->bound_auto && ctx->slabs_state >= 0 && max_offtargets > 0) {
        ctx->bound_mode = 1;
        rc = ffh_scan_bounded(ctx, guides, n_guides, max_mismatch, max_offtargets);
    }
    return rc;
}
static ffh_result *merge_results(ffh_ctx *ctx, const ffh_result *a, const ffh_result *b, unsigned flags) {
    const bool lists = !(flags & FFH_FINALIZE_SUMMARIES_ONLY), want_pos = lists && !(flags & FFH_FINALIZE_NO_POSITIONS), want_cfd = lists && !(flags & FFH_FINALIZE_NO_HIT_SCORES);
    const uint32_t Ga = a->n_guides, Gb = b->n_guides;
    const uint64_t Ha = lists ? a->n_hits : 0, Hb = lists ? b->n_hits : 0;
    ffh_result *r = new (std::nothrow) ffh_result();
    if (!r || !r->allocate(ctx->pool, Ga + Gb, Ha + Hb, lists, want_cfd, want_pos) || (want_pos && !r->allocate_positions(a->n_positions + b->n_positions))) { delete r; return nullptr; }
    r->scores_valid = a->scores_valid;
    if (Ga) std::memcpy(r->summaries, a->summaries, (size_t)Ga * sizeof(ffh_guide_summary));
    if (Gb) std::memcpy(r->summaries + Ga, b->summaries, (size_t)Gb * sizeof(ffh_guide_summary));
    if (!lists) { r->offsets_pending = true; return r; }
    std::memcpy(r->guide_offsets, a->guide_offsets, ((size_t)Ga + 1) * 8);
    for (uint32_t g = 0; g <= Gb; ++g) r->guide_offsets[Ga + g] = Ha + b->guide_offsets[g];
    auto cat = [&](auto *dst, const auto *pa, const auto *pb) {
        if (!dst) return;
        if (Ha) std::memcpy(dst, pa, (size_t)Ha * sizeof(*dst));
        if (Hb) std::memcpy(dst + Ha, pb, (size_t)Hb * sizeof(*dst));
    };
    cat(r->hit_targets, a->hit_targets, b->hit_targets);
    cat(r->hit_mm, a->hit_mm, b->hit_mm);
    if (want_cfd) cat(r->hit_cfd, a->hit_cfd, b->hit_cfd);
    if (want_pos) {
        r->pos_offsets_pending = true;   // folded from the counts in the hit target longs when first asked for (settle_pos_offsets)
        if (a->n_positions) std::memcpy(r->positions, a->positions, (size_t)a->n_positions * 8);
        if (b->n_positions) std::memcpy(r->positions + a->n_positions, b->positions, (size_t)b->n_positions * 8);
    }
    return r;
}
// The list-delivering discover of a large guide set against a large database, in two halves: the first half's lists cross the link while
// the second half is scanned (finalize_lists).  Returns 1 when the pipeline was abandoned (a scan that must be split further, a second
// half that does not fit the block the first one sized): nothing is left in flight and the caller runs the unsplit call.
static int discover_pipelined(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets, unsigned flags, ffh_result **out) {
    // (60 : 40 -- the second part's scan should take about as long as the first part's lists need on the link, and a scan of
    // fewer guides costs more per guide: profiles/r05/ab_log.txt 7)
    const uint32_t ga = (uint32_t)((uint64_t)n_guides * 3 / 5);
    ListPipe lp;
    lp.G_total = n_guides; lp.pipelined = true; lp.last = false;
    auto abandon = [&]() { (void)hipStreamSynchronize(ctx->st); (void)hipStreamSynchronize(ctx->copy_st); delete lp.r; lp.r = nullptr; return 1; };
    int rc = scan_retry_bounded(ctx, guides, ga, max_mismatch, max_offtargets);
    if (rc) return ctx->too_many_hits ? 1 : rc;
    rc = finalize_lists(ctx, nullptr, max_offtargets, flags, lp);
    if (rc) { (void)abandon(); return rc; }
    lp.g_base = ga; lp.h_base = lp.Hr; lp.p_base = lp.Pr; lp.last = true;
    ctx->pg_slot = 1;
    rc = scan_retry_bounded(ctx, guides + ga, n_guides - ga, max_mismatch, max_offtargets);
    ctx->pg_slot = 0;
    if (rc) { const bool again = ctx->too_many_hits; (void)abandon(); return again ? 1 : rc; }
    rc = finalize_lists(ctx, nullptr, max_offtargets, flags, lp);
    if (rc) { (void)abandon(); return rc; }
    if (lp.overflow) return abandon();
    *out = lp.r;
    return FFH_OK;
}
static int discover_split(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets, unsigned flags, ffh_result **out) {
    if (ctx && !(flags & FFH_FINALIZE_SUMMARIES_ONLY) && n_guides >= 2 && ctx->sw.pipeline) {   // (FFH_PIPELINE=1 only: see ffh_debug.hpp)
        const int rc = discover_pipelined(ctx, guides, n_guides, max_mismatch, max_offtargets, flags, out);
        if (rc <= 0) return rc;   // (1: abandoned, go on unsplit)
    }
    int rc = scan_retry_bounded(ctx, guides, n_guides, max_mismatch, max_offtargets);
    if (!rc) return ffh_finalize(ctx, nullptr, max_offtargets, flags, out);
    if (!ctx || !ctx->too_many_hits || n_guides < 2) return rc;
    const uint32_t h = n_guides / 2;
    ffh_result *a = nullptr, *b = nullptr;
    rc = discover_split(ctx, guides, h, max_mismatch, max_offtargets, flags, &a);
    if (!rc) rc = discover_split(ctx, guides + h, n_guides - h, max_mismatch, max_offtargets, flags, &b);
    if (!rc) {
        *out = merge_results(ctx, a, b, flags);
        if (!*out) { ctx->err = "out of (pinned) host memory"; rc = FFH_E_NOMEM; }
        else ctx->err.clear();
    }
    delete a; delete b;
    return rc;
}

int ffh_discover(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets, unsigned flags, ffh_result **out) {
    if (max_offtargets < 0 || !out) { if (ctx) ctx->err = "bad argument"; return FFH_E_ARG; }
    return discover_split(ctx, guides, n_guides, max_mismatch, max_offtargets, flags, out);
}

int ffh_score_lists(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, const uint64_t *guide_offsets, const uint64_t *hit_targets, ffh_result **out) {
    if (!ctx || !out || (n_guides && (!guides || !guide_offsets))) { if (ctx) ctx->err = "bad argument"; return FFH_E_ARG; }
    if (ctx->enzyme == 0) { ctx->err = "the context has no enzyme yet"; return FFH_E_STATE; }
    FFH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->st;
    const uint32_t G = n_guides;
    const uint64_t H = G ? guide_offsets[G] : 0;
    if (H && !hit_targets) { ctx->err = "bad argument"; return FFH_E_ARG; }
    std::vector<uint32_t> hit_guide((size_t)H), n_ret(G), ot(G), full(G, 0u);
    for (uint32_t g = 0; g < G; ++g) {
        if (guide_offsets[g + 1] < guide_offsets[g] || guide_offsets[g + 1] > H) { ctx->err = "guide_offsets must be non-decreasing"; return FFH_E_ARG; }
        uint64_t tot = 0;
        for (uint64_t h = guide_offsets[g]; h < guide_offsets[g + 1]; ++h) { hit_guide[(size_t)h] = g; tot += hit_targets[h] >> 48; }
        n_ret[g] = (uint32_t)(guide_offsets[g + 1] - guide_offsets[g]);
        ot[g] = (uint32_t)std::min<uint64_t>(tot, 0xFFFFFFFFull);
    }
    ctx->scanned = false;  // the scratch arrays of a previous scan are reused below
    FFH_HIP(ctx->guides.reserve((size_t)G + 1));
    FFH_HIP(ctx->n_ret.reserve((size_t)G + 1));
    FFH_HIP(ctx->ot_count.reserve((size_t)G + 1));
    FFH_HIP(ctx->full.reserve((size_t)G + 1));
    FFH_HIP(ctx->ret_off.reserve((size_t)G + 2));
    FFH_HIP(ctx->summ.reserve((size_t)G + 1));
    FFH_HIP(ctx->out_target.reserve(H + 1));
    FFH_HIP(ctx->out_tidx.reserve(H + 1));
    FFH_HIP(ctx->out_mm.reserve(H + 1));
    FFH_HIP(ctx->out_cnt.reserve(H + 1));
    FFH_HIP(ctx->out_cfd.reserve(H + 1));
    FFH_HIP(ctx->out_hsu.reserve(H + 1));
    FFH_HIP(ctx->out_jost.reserve(H + 1));
    if (G) {
        FFH_HIP(hipMemcpyAsync(ctx->guides.p, guides, (size_t)G * 8, hipMemcpyHostToDevice, st));
        FFH_HIP(hipMemcpyAsync(ctx->n_ret.p, n_ret.data(), (size_t)G * 4, hipMemcpyHostToDevice, st));
        FFH_HIP(hipMemcpyAsync(ctx->ot_count.p, ot.data(), (size_t)G * 4, hipMemcpyHostToDevice, st));
        FFH_HIP(hipMemcpyAsync(ctx->full.p, full.data(), (size_t)G * 4, hipMemcpyHostToDevice, st));
        FFH_HIP(hipMemcpyAsync(ctx->ret_off.p, guide_offsets, ((size_t)G + 1) * 8, hipMemcpyHostToDevice, st));
    }
    if (H) {
        FFH_HIP(hipMemcpyAsync(ctx->out_target.p, hit_targets, H * 8, hipMemcpyHostToDevice, st));
        FFH_HIP(hipMemcpyAsync(ctx->out_tidx.p, hit_guide.data(), H * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_score_list, dim3(blocks_for(H, 256)), dim3(256), 0, st, ctx->out_target.p, ctx->out_tidx.p, H, ctx->guides.p, ctx->geo, ctx->d_tab,
                           ctx->out_mm.p, ctx->out_cnt.p, ctx->out_cfd.p, ctx->out_hsu.p, ctx->out_jost.p);
    }
    if (G) hipLaunchKernelGGL(k_guide_aggregate, dim3(blocks_for(G, 4)), dim3(256), 0, st, ctx->ret_off.p, ctx->n_ret.p, ctx->ot_count.p, ctx->full.p, ctx->out_mm.p,
                              ctx->out_cnt.p, ctx->out_cfd.p, ctx->out_hsu.p, (const double *)ctx->out_jost.p, G, ctx->summ.p);
    FFH_HIP(hipGetLastError());
    ffh_result *r = new (std::nothrow) ffh_result();
    if (!r || !r->allocate(ctx->pool, G, H, true)) { delete r; ctx->err = "out of (pinned) host memory"; return FFH_E_NOMEM; }
    r->scores_valid = ctx->geo.cas9_23;
    if (G) std::memcpy(r->guide_offsets, guide_offsets, ((size_t)G + 1) * 8);
    else r->guide_offsets[0] = 0;
    if (H) std::memcpy(r->hit_targets, hit_targets, (size_t)H * 8);
    std::memset(r->pos_offsets, 0, ((size_t)H + 1) * 8);
    hipError_t e = hipSuccess;
    if (G) e = hipMemcpyAsync(r->summaries, ctx->summ.p, (size_t)G * sizeof(ffh_guide_summary), hipMemcpyDeviceToHost, st);
    if (H && e == hipSuccess) e = hipMemcpyAsync(r->hit_mm, ctx->out_mm.p, H, hipMemcpyDeviceToHost, st);
    if (H && e == hipSuccess) e = hipMemcpyAsync(r->hit_cfd, ctx->out_cfd.p, H * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { ctx->err = std::string("result copy: ") + hipGetErrorString(e); delete r; return FFH_E_HIP; }
    *out = r;
    return FFH_OK;
}

int ffh_get_timings(const ffh_ctx *ctx, ffh_timings *out) {
    if (!ctx || !out) return FFH_E_ARG;
    finish_scan_timings(const_cast<ffh_ctx *>(ctx));
    finish_finalize_timing(const_cast<ffh_ctx *>(ctx));
    *out = ctx->tm;
    return FFH_OK;
}

uint32_t ffh_result_n_guides(const ffh_result *r) { return r->n_guides; }
static void settle_offsets(const ffh_result *cr) {
    ffh_result *r = const_cast<ffh_result *>(cr);
    if (!r->offsets_pending) return;
    std::call_once(r->offsets_once, [r] {
        uint64_t run = 0;
        for (uint32_t g = 0; g < r->n_guides; ++g) { r->guide_offsets[g] = run; run += r->summaries[g].n_hits; }
        r->guide_offsets[r->n_guides] = run;
        r->n_hits = run;
    });
}
uint64_t ffh_result_n_hits(const ffh_result *r) { settle_offsets(r); return r->n_hits; }
uint64_t ffh_result_n_positions(const ffh_result *r) { return r->n_positions; }
int ffh_result_scores_valid(const ffh_result *r) { return r->scores_valid; }
const ffh_guide_summary *ffh_result_summaries(const ffh_result *r) { return r->summaries; }
const uint64_t *ffh_result_guide_offsets(const ffh_result *r) { settle_offsets(r); return r->guide_offsets; }
const uint64_t *ffh_result_hit_targets(const ffh_result *r) { return r->hit_targets; }
const uint8_t *ffh_result_hit_mismatches(const ffh_result *r) { return r->hit_mm; }
const double *ffh_result_hit_cfd(const ffh_result *r) { return r->hit_cfd; }
// exclusive prefix sums of the hits' position counts (bits 63:48 of the target longs), a few host threads over contiguous slices
static void settle_pos_offsets(const ffh_result *cr) {
    ffh_result *r = const_cast<ffh_result *>(cr);
    if (!r->pos_offsets_pending) return;
    // (once, whoever comes first; a second thread waits here until the offsets are complete -- a plain flag let it read them half
    // written: "cannot create std::vector larger than max_size()" from the CLI's parallel row formatting)
    std::call_once(r->pos_offsets_once, [r] {
        const uint64_t H = r->n_hits;
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned nt = (unsigned)std::min<uint64_t>(std::min(8u, hw), H / 262144 + 1);
        std::vector<uint64_t> part(nt + 1, 0);
        auto slice = [&](unsigned t, uint64_t &a, uint64_t &b) { a = H * t / nt; b = H * (t + 1) / nt; };
        auto run = [&](auto &&fn) {
            if (nt == 1) { fn(0u); return; }
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(fn, t);
            for (auto &x : th) x.join();
        };
        run([&](unsigned t) { uint64_t a, b, s = 0; slice(t, a, b); for (uint64_t h = a; h < b; ++h) s += r->hit_targets[h] >> 48; part[t + 1] = s; });
        for (unsigned t = 0; t < nt; ++t) part[t + 1] += part[t];
        run([&](unsigned t) { uint64_t a, b, s = part[t]; slice(t, a, b); for (uint64_t h = a; h < b; ++h) { r->pos_offsets[h] = s; s += r->hit_targets[h] >> 48; } });
        r->pos_offsets[H] = part[nt];
    });
}
const uint64_t *ffh_result_pos_offsets(const ffh_result *r) { if (r->pos_offsets) settle_pos_offsets(r); return r->pos_offsets; }
const uint64_t *ffh_result_positions(const ffh_result *r) { return r->positions; }
void ffh_result_free(ffh_result *r) { delete r; }

}  // extern "C"

// =====================================================================================================================
// index on the device (ffh_index.hpp): sites -> sort -> unique targets + position lists -> ffh_db_write
// =====================================================================================================================
#include "ffh_index.hpp"

struct ffh_indexer {
    int device = 0, enzyme = 0;
    hipStream_t st = nullptr;
    SitePattern pat{};
    std::string err;
    std::vector<std::string> contigs;
    DevBuf<uint8_t> seq;
    DevBuf<uint32_t> blk_cnt;
    DevBuf<uint64_t> blk_off, scan_tmp;
    DevBuf<uint64_t> keys, pos;  // sites in discovery order
    uint64_t n_sites = 0, n_bases = 0;
    double scan_ms = 0;
};

static void site_pattern(int enzyme, SitePattern &p) {  // fwdRegex / revRegex, standards/StandardScanParameters.scala:104-211
    const uint8_t A = 1, Cc = 2, G = 4, T = 8, N = 15;
    const int L = enzyme == 1 ? 24 : (enzyme >= 5 ? 22 : 23);
    p.len = L;
    for (int k = 0; k < 24; ++k) { p.fwd[k] = 0; p.rev[k] = 0; }
    for (int k = 0; k < L; ++k) { p.fwd[k] = N; p.rev[k] = N; }
    switch (enzyme) {
        case 1: p.fwd[0] = p.fwd[1] = p.fwd[2] = T; p.rev[L - 1] = p.rev[L - 2] = p.rev[L - 3] = A; break;      // TTTN...  /  ...NAAA (:209-211)
        case 2: case 5: p.fwd[L - 2] = A | G; p.fwd[L - 1] = G; p.rev[0] = Cc; p.rev[1] = Cc | T; break;          // N[AG]G   /  C[CT]N (:104-106, :126-128)
        case 3: case 6: p.fwd[L - 2] = G; p.fwd[L - 1] = G; p.rev[0] = Cc; p.rev[1] = Cc; break;                  // NGG      /  CCN   (:148-150, :170-172)
        case 4: p.fwd[L - 2] = A; p.fwd[L - 1] = G; p.rev[0] = Cc; p.rev[1] = T; break;                            // NAG      /  CTN   (:192-194)
    }
}

extern "C" {

ffh_indexer *ffh_indexer_create(int device_id, int enzyme_index) {
    if (enzyme_index < 1 || enzyme_index > 6) { g_create_error = "Unable to find the correct parameter pack for enzyme: " + std::to_string(enzyme_index); return nullptr; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_create_error = "no HIP device available (flashfry_hip has no CPU fallback)"; return nullptr; }
    if (device_id < 0 || device_id >= n) { g_create_error = "device id out of range"; return nullptr; }
    ffh_indexer *ix = new (std::nothrow) ffh_indexer();
    if (!ix) { g_create_error = "out of memory"; return nullptr; }
    ix->device = device_id; ix->enzyme = enzyme_index;
    site_pattern(enzyme_index, ix->pat);
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->st, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_error = std::string("HIP initialisation failed: ") + hipGetErrorString(e); delete ix; return nullptr; }
    return ix;
}

void ffh_indexer_destroy(ffh_indexer *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->st) (void)hipStreamSynchronize(ix->st);
    if (ix->st) (void)hipStreamDestroy(ix->st);
    delete ix;
}

const char *ffh_indexer_last_error(const ffh_indexer *ix) { return ix ? ix->err.c_str() : g_create_error.c_str(); }

int ffh_indexer_add_contig(ffh_indexer *ctx, const char *name, const char *sequence, uint64_t length) {
    if (!ctx || !name || (length && !sequence)) { if (ctx) ctx->err = "null argument"; return FFH_E_ARG; }
    if (length >= (1ull << 32)) { ctx->err = "contig longer than 2^32 bases: positions are 32-bit (BitPosition.scala:51-63)"; return FFH_E_ARG; }
    if (ctx->contigs.size() >= (1u << 20) - 1) { ctx->err = "more than 2^20 contigs (BitPosition.scala:51-63)"; return FFH_E_ARG; }
    ctx->contigs.emplace_back(name);
    const uint32_t contig_id = (uint32_t)ctx->contigs.size();  // BitPosition.addReference: ids from 1 in order of appearance
    ctx->n_bases += length;
    if (length < (uint64_t)ctx->pat.len) return FFH_OK;
    FFH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->st;
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t nb = (uint32_t)((length + kSiteTile - 1) / kSiteTile);
    FFH_HIP(ctx->seq.reserve(length + 64));
    FFH_HIP(ctx->blk_cnt.reserve(2 * (size_t)nb + 8));
    FFH_HIP(ctx->blk_off.reserve(2 * (size_t)nb + 8));
    FFH_HIP(ctx->scan_tmp.reserve(scan_scratch_elems_safe(2 * (uint64_t)nb)));
    FFH_HIP(hipMemcpyAsync(ctx->seq.p, sequence, length, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_site_scan<false>, dim3(nb), dim3(kSiteThreads), 0, st, ctx->seq.p, length, ctx->pat, contig_id, nb, ctx->blk_cnt.p,
                       (const uint64_t *)nullptr, 0ull, (uint64_t *)nullptr, (uint64_t *)nullptr);
    exclusive_scan<uint32_t, uint64_t>(ctx->blk_cnt.p, 2 * (uint64_t)nb, ctx->blk_off.p, ctx->scan_tmp.p, st);
    uint64_t found = 0;
    FFH_HIP(hipMemcpyAsync(&found, ctx->blk_off.p + 2 * (size_t)nb, 8, hipMemcpyDeviceToHost, st));
    FFH_HIP(hipStreamSynchronize(st));
    if (ctx->n_sites + found >= (1ull << 32) - 64) { ctx->err = "more than 2^32 target sites"; return FFH_E_ARG; }
    if (found) {
        FFH_HIP(grow_keep(ctx->keys, ctx->n_sites, ctx->n_sites + found, st));
        FFH_HIP(grow_keep(ctx->pos, ctx->n_sites, ctx->n_sites + found, st));
        hipLaunchKernelGGL(k_site_scan<true>, dim3(nb), dim3(kSiteThreads), 0, st, ctx->seq.p, length, ctx->pat, contig_id, nb, ctx->blk_cnt.p,
                           (const uint64_t *)ctx->blk_off.p, ctx->n_sites, ctx->keys.p, ctx->pos.p);
        FFH_HIP(hipGetLastError());
        FFH_HIP(hipStreamSynchronize(st));
        ctx->n_sites += found;
    }
    ctx->scan_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return FFH_OK;
}

int ffh_indexer_finish(ffh_indexer *ctx, const char *db_path, int bin_width, ffh_index_stats *stats) {
    if (!ctx || !db_path) { if (ctx) ctx->err = "null argument"; return FFH_E_ARG; }
    FFH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->st;
    const uint64_t S = ctx->n_sites;
    const auto t0 = std::chrono::steady_clock::now();
    DevBuf<uint64_t> alt_k, alt_v, d_targets, d_positions, d_posoff, scr64;
    DevBuf<uint32_t> table, offs, scr32, head, rank, start, count;
    std::vector<uint64_t> h_targets, h_positions;
    uint64_t n_targets = 0, n_positions = 0;
    if (S) {
        // stable sort of (sequence, position) by sequence: CRISPRSite.compare = the bases (crispr/CRISPRSite.scala:44)
        const uint32_t nb = sort_nblocks(S);
        FFH_HIP(alt_k.reserve(S)); FFH_HIP(alt_v.reserve(S));
        FFH_HIP(table.reserve((size_t)kSortTableDigits * nb + 8)); FFH_HIP(offs.reserve((size_t)kSortTableDigits * nb + 8));
        FFH_HIP(scr32.reserve(scan_scratch_elems_safe(std::max<uint64_t>((uint64_t)kSortTableDigits * nb, S + 1))));
        SortScratch ss;
        ss.alt = alt_k.p; ss.val_alt = alt_v.p; ss.table = table.p; ss.offs = offs.p; ss.scan_tmp = scr32.p;
        uint64_t *sk = nullptr, *sv = nullptr;
        radix_sort_pairs(ctx->keys.p, ctx->pos.p, S, 0, 2 * ctx->pat.len, ss, st, sk, sv);
        // runs of equal sequences -> one target with its (capped) count and position list
        FFH_HIP(head.reserve(S + 8)); FFH_HIP(rank.reserve(S + 8));
        hipLaunchKernelGGL(k_run_heads, dim3(blocks_for(S, 256)), dim3(256), 0, st, sk, S, head.p);
        exclusive_scan<uint32_t, uint32_t>(head.p, S, rank.p, scr32.p, st);
        uint32_t runs = 0;
        FFH_HIP(hipMemcpyAsync(&runs, rank.p + S, 4, hipMemcpyDeviceToHost, st));
        FFH_HIP(hipStreamSynchronize(st));
        n_targets = runs;
        FFH_HIP(start.reserve((size_t)runs + 8)); FFH_HIP(count.reserve((size_t)runs + 8));
        FFH_HIP(d_targets.reserve((size_t)runs + 1)); FFH_HIP(d_posoff.reserve((size_t)runs + 2));
        FFH_HIP(scr64.reserve(scan_scratch_elems_safe(runs)));
        hipLaunchKernelGGL(k_run_starts, dim3(blocks_for(S, 256)), dim3(256), 0, st, head.p, rank.p, S, runs, start.p);
        hipLaunchKernelGGL(k_run_targets, dim3(blocks_for(runs, 256)), dim3(256), 0, st, sk, start.p, runs, d_targets.p, count.p);
        exclusive_scan<uint32_t, uint64_t>(count.p, runs, d_posoff.p, scr64.p, st);
        FFH_HIP(hipMemcpyAsync(&n_positions, d_posoff.p + runs, 8, hipMemcpyDeviceToHost, st));
        FFH_HIP(hipStreamSynchronize(st));
        FFH_HIP(d_positions.reserve(n_positions + 1));
        hipLaunchKernelGGL(k_run_positions, dim3(blocks_for(S, 256)), dim3(256), 0, st, sv, head.p, rank.p, start.p, d_posoff.p, S, d_positions.p);
        FFH_HIP(hipGetLastError());
        try { h_targets.resize(n_targets); h_positions.resize(n_positions); } catch (const std::bad_alloc &) { ctx->err = "out of host memory"; return FFH_E_NOMEM; }
        FFH_HIP(hipMemcpyAsync(h_targets.data(), d_targets.p, n_targets * 8, hipMemcpyDeviceToHost, st));
        if (n_positions) FFH_HIP(hipMemcpyAsync(h_positions.data(), d_positions.p, n_positions * 8, hipMemcpyDeviceToHost, st));
        FFH_HIP(hipStreamSynchronize(st));
    }
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<const char *> names;
    for (const auto &c : ctx->contigs) names.push_back(c.c_str());
    const int rc = ffh_db_write(db_path, ctx->enzyme, bin_width, names.data(), (uint32_t)names.size(), h_targets.data(), n_targets, h_positions.data(), n_positions);
    if (rc) { ctx->err = g_create_error; return rc; }
    if (stats) {
        stats->n_bases = ctx->n_bases; stats->n_sites = S; stats->n_targets = n_targets; stats->n_positions = n_positions; stats->n_contigs = (uint32_t)ctx->contigs.size();
        stats->scan_ms = ctx->scan_ms;
        stats->sort_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        stats->write_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    }
    return FFH_OK;
}

}  // extern "C"

// =====================================================================================================================
// config C5: mismatches + one bulge, Cas12a (ffh_bulge.hpp)
// =====================================================================================================================
#include "ffh_bulge.hpp"

struct ffh_bulge_result {
    uint32_t n_guides = 0;
    std::vector<uint64_t> guide_offsets, hit_targets;
    std::vector<uint8_t> hit_mm, hit_type, hit_pos;
};

extern "C" {

static int bulge_impl(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_bulge, unsigned flags, ffh_bulge_result **out);
// (more candidate records than one search holds: the guide set is halved and the halves' results are concatenated, as ffh_discover does)
int ffh_discover_bulge(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_bulge, unsigned flags, ffh_bulge_result **out) {
    if (!ctx || !out) { if (ctx) ctx->err = "bad argument"; return FFH_E_ARG; }
    ctx->too_many_hits = false;
    int rc = bulge_impl(ctx, guides, n_guides, max_mismatch, max_bulge, flags, out);
    if (!rc || !ctx->too_many_hits || n_guides < 2) return rc;
    const uint32_t h = n_guides / 2;
    ffh_bulge_result *a = nullptr, *b = nullptr;
    rc = ffh_discover_bulge(ctx, guides, h, max_mismatch, max_bulge, flags, &a);
    if (!rc) rc = ffh_discover_bulge(ctx, guides + h, n_guides - h, max_mismatch, max_bulge, flags, &b);
    if (!rc) {
        try {
            const uint64_t Ha = a->hit_targets.size();
            a->n_guides = n_guides;
            a->guide_offsets.reserve((size_t)n_guides + 1);
            for (uint32_t g = 1; g <= n_guides - h; ++g) a->guide_offsets.push_back(Ha + b->guide_offsets[g]);
            a->hit_targets.insert(a->hit_targets.end(), b->hit_targets.begin(), b->hit_targets.end());
            a->hit_mm.insert(a->hit_mm.end(), b->hit_mm.begin(), b->hit_mm.end());
            a->hit_type.insert(a->hit_type.end(), b->hit_type.begin(), b->hit_type.end());
            a->hit_pos.insert(a->hit_pos.end(), b->hit_pos.begin(), b->hit_pos.end());
            *out = a; a = nullptr;
            ctx->err.clear();
        } catch (const std::bad_alloc &) { ctx->err = "out of host memory"; rc = FFH_E_NOMEM; }
    }
    delete a; delete b;
    return rc;
}
static int bulge_impl(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_bulge, unsigned flags, ffh_bulge_result **out) {
    if (!ctx || !out || (n_guides && !guides) || max_mismatch < 0 || max_bulge < 0 || max_bulge > 1) { if (ctx) ctx->err = "bad argument"; return FFH_E_ARG; }
    if (ctx->img[0].width < 0) { ctx->err = "no database loaded"; return FFH_E_STATE; }
    if (ctx->enzyme != 1) { ctx->err = "the bulge search is specified for Cas12a / Cpf1 (enzyme index 1) only"; return FFH_E_ARG; }
    if (n_guides >= (1u << 24)) { ctx->err = "too many guides for one bulge search"; return FFH_E_ARG; }
    FFH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->st;
    std::unique_ptr<ffh_bulge_result> r(new (std::nothrow) ffh_bulge_result());
    if (!r) { ctx->err = "out of memory"; return FFH_E_NOMEM; }
    r->n_guides = n_guides;
    r->guide_offsets.assign((size_t)n_guides + 1, 0);
    const uint64_t T = ctx->T;
    int tbits = 1;
    while (tbits < 32 && (1ull << tbits) < std::max<uint64_t>(T, 2)) ++tbits;
    int gbits = 1;
    while ((1u << gbits) < std::max<uint32_t>(n_guides, 2)) ++gbits;
    uint64_t n_hits = 0;
    DevBuf<uint64_t> d_guides, key, val, alt_k, alt_v, d_target, d_dst, d_key, scr64;
    DevBuf<uint32_t> table, offs, scr32, d_flag, d_pat[3];
    DevBuf<uint8_t> d_mm, d_type, d_pos;
    const bool brute = (flags & FFH_BULGE_BRUTE_FORCE) != 0;
    if (n_guides && T) {
        FFH_HIP(d_guides.reserve(n_guides));
        FFH_HIP(hipMemcpyAsync(d_guides.p, guides, (size_t)n_guides * 8, hipMemcpyHostToDevice, st));
        // the seeds of the candidate search (ffh_bulge.hpp): P on the prefix image, D and R on the suffix image
        BulgeSeed seeds[3];
        int n_seeds = 0;
        if (!brute) {
            const int a = ctx->img[0].width, sfx = ctx->img[1].width;
            for (int kind = 0; kind < (max_bulge ? 3 : 1); ++kind) {
                const Image &im = ctx->img[kind == 0 ? 0 : 1];
                if (im.direct) { ctx->err = "internal: direct image in a bulge search"; return FFH_E_STATE; }   // (Cpf1 has a 5' PAM: never direct)
                const int w = kind == 0 ? a : sfx;
                if (kind == 2 && w == 0) continue;  // one suffix bucket: seed D already visits it
                std::vector<uint32_t> pat;
                if (kind < 2) pat = patterns_for(ctx, w, max_mismatch);
                else {  // R: max_mismatch substitutions over the s - 1 paired bases (bucket positions 1 .. s-1), any base at position 0
                    const int n = w - 1;
                    for (uint32_t q : patterns_for(ctx, n, max_mismatch)) {
                        const uint32_t lo = q & ((1u << n) - 1u), hi = q >> n;
                        for (uint32_t d = 0; d < 4; ++d) pat.push_back(((hi << 1 | (d >> 1)) << w) | (lo << 1 | (d & 1u)));
                    }
                }
                FFH_HIP(d_pat[kind].reserve(pat.size()));
                FFH_HIP(hipMemcpyAsync(d_pat[kind].p, pat.data(), pat.size() * 4, hipMemcpyHostToDevice, st));
                FFH_HIP(hipStreamSynchronize(st));  // pat is a local
                BulgeSeed &S = seeds[n_seeds++];
                S.bstart = im.bstart.p; S.gstart = im.gstart.p; S.gwords = im.gwords.p; S.tidx = im.tidx.p; S.patterns = d_pat[kind].p;
                S.n_pat = (uint32_t)pat.size(); S.width = w; S.rest = im.rest; S.gw = (int)group_words((uint32_t)im.rest); S.kind = kind;
            }
        }
        unsigned long long *cursor = ctx->d_counters + 12;
        size_t cap = std::max<size_t>(1u << 20, (size_t)n_guides * 1024);
        for (;;) {
            FFH_HIP(key.reserve(cap)); FFH_HIP(val.reserve(cap));
            cap = std::min(key.cap, val.cap);
            FFH_HIP(hipMemsetAsync(cursor, 0, 8, st));
            if (brute)
                hipLaunchKernelGGL(k_bulge_scan, dim3(blocks_for(T, 256)), dim3(256), 0, st, ctx->targets.p, T, d_guides.p, n_guides, ctx->geo, max_mismatch, max_bulge,
                                   (flags & FFH_BULGE_PAM_TTTV) ? 1 : 0, tbits, key.p, val.p, cursor, (uint64_t)cap);
            else
                for (int i = 0; i < n_seeds; ++i) {
                    // one wave per (guide, 64 patterns); a launch holds at most 2^22 blocks of four waves (HIP refuses grids of 2^32
                    // threads or more: ADVICE r2), so a large guide set goes in several launches
                    const uint64_t slices = (seeds[i].n_pat + 63) / 64;
                    const uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_guides, ((1ull << 24) - 4) / slices));
                    for (uint32_t g0 = 0; g0 < n_guides; g0 += per) {
                        const uint32_t ng = std::min(per, n_guides - g0);
                        const uint64_t waves = (uint64_t)ng * slices;
                        hipLaunchKernelGGL(k_bulge_seed, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, seeds[i], (const uint64_t *)ctx->targets.p,
                                           (const uint64_t *)d_guides.p + g0, ng, g0, ctx->geo, max_mismatch, max_bulge, (flags & FFH_BULGE_PAM_TTTV) ? 1 : 0, tbits, key.p, val.p,
                                           cursor, (uint64_t)cap);
                    }
                }
            FFH_HIP(hipGetLastError());
            unsigned long long found = 0;
            FFH_HIP(hipMemcpyAsync(&found, cursor, 8, hipMemcpyDeviceToHost, st));
            FFH_HIP(hipStreamSynchronize(st));
            if (found >= ctx->sw.raw_hit_limit) {   // (sort offsets are 32-bit: ffh_discover_bulge splits the guide set)
                ctx->too_many_hits = true;
                ctx->err = "more than 2^32 bulge candidate records in one search";
                return FFH_E_ARG;
            }
            if (found <= cap) { n_hits = found; break; }
            cap = (size_t)(found + found / 8);  // the buffer was too small: grow to what the scan found and run it again
        }
    }
    if (n_hits) {
        const uint32_t nb = sort_nblocks(n_hits);
        FFH_HIP(alt_k.reserve(n_hits)); FFH_HIP(alt_v.reserve(n_hits));
        FFH_HIP(table.reserve((size_t)kSortTableDigits * nb + 8)); FFH_HIP(offs.reserve((size_t)kSortTableDigits * nb + 8));
        FFH_HIP(scr32.reserve(scan_scratch_elems_safe((uint64_t)kSortTableDigits * nb)));
        SortScratch ss;
        ss.alt = alt_k.p; ss.val_alt = alt_v.p; ss.table = table.p; ss.offs = offs.p; ss.scan_tmp = scr32.p;
        uint64_t *sk = nullptr, *sv = nullptr;
        radix_sort_pairs(key.p, val.p, n_hits, 0, tbits + gbits, ss, st, sk, sv);  // (guide, database order)
        // a pair reached through two seeds appears twice with the same record: keep the first of every key
        FFH_HIP(d_flag.reserve(n_hits + 1)); FFH_HIP(d_dst.reserve(n_hits + 2)); FFH_HIP(scr64.reserve(scan_scratch_elems_safe(n_hits + 1)));
        hipLaunchKernelGGL(k_bulge_flag_unique, dim3(blocks_for(n_hits, 256)), dim3(256), 0, st, (const uint64_t *)sk, n_hits, d_flag.p);
        exclusive_scan<uint32_t, uint64_t>(d_flag.p, n_hits, d_dst.p, scr64.p, st);
        uint64_t n_unique = 0;
        FFH_HIP(hipMemcpyAsync(&n_unique, d_dst.p + n_hits, 8, hipMemcpyDeviceToHost, st));
        FFH_HIP(hipStreamSynchronize(st));
        FFH_HIP(d_key.reserve(n_unique)); FFH_HIP(d_target.reserve(n_unique)); FFH_HIP(d_mm.reserve(n_unique)); FFH_HIP(d_type.reserve(n_unique)); FFH_HIP(d_pos.reserve(n_unique));
        hipLaunchKernelGGL(k_bulge_unpack, dim3(blocks_for(n_hits, 256)), dim3(256), 0, st, (const uint64_t *)sk, (const uint64_t *)sv, n_hits, tbits, (const uint64_t *)ctx->targets.p,
                           (const uint32_t *)d_flag.p, (const uint64_t *)d_dst.p, d_key.p, d_target.p, d_mm.p, d_type.p, d_pos.p);
        FFH_HIP(hipGetLastError());
        std::vector<uint64_t> keys;
        try {
            keys.resize(n_unique);
            r->hit_targets.resize(n_unique); r->hit_mm.resize(n_unique); r->hit_type.resize(n_unique); r->hit_pos.resize(n_unique);
        } catch (const std::bad_alloc &) { ctx->err = "out of host memory"; return FFH_E_NOMEM; }
        FFH_HIP(hipMemcpyAsync(keys.data(), d_key.p, n_unique * 8, hipMemcpyDeviceToHost, st));
        FFH_HIP(hipMemcpyAsync(r->hit_targets.data(), d_target.p, n_unique * 8, hipMemcpyDeviceToHost, st));
        FFH_HIP(hipMemcpyAsync(r->hit_mm.data(), d_mm.p, n_unique, hipMemcpyDeviceToHost, st));
        FFH_HIP(hipMemcpyAsync(r->hit_type.data(), d_type.p, n_unique, hipMemcpyDeviceToHost, st));
        FFH_HIP(hipMemcpyAsync(r->hit_pos.data(), d_pos.p, n_unique, hipMemcpyDeviceToHost, st));
        FFH_HIP(hipStreamSynchronize(st));
        for (uint64_t i = 0; i < n_unique; ++i) ++r->guide_offsets[(size_t)(keys[i] >> tbits) + 1];
        for (uint32_t g = 0; g < n_guides; ++g) r->guide_offsets[g + 1] += r->guide_offsets[g];
    }
    *out = r.release();
    return FFH_OK;
}

uint32_t ffh_bulge_result_n_guides(const ffh_bulge_result *r) { return r ? r->n_guides : 0; }
uint64_t ffh_bulge_result_n_hits(const ffh_bulge_result *r) { return r ? (uint64_t)r->hit_targets.size() : 0; }
const uint64_t *ffh_bulge_result_guide_offsets(const ffh_bulge_result *r) { return r ? r->guide_offsets.data() : nullptr; }
const uint64_t *ffh_bulge_result_hit_targets(const ffh_bulge_result *r) { return r ? r->hit_targets.data() : nullptr; }
const uint8_t *ffh_bulge_result_hit_mismatches(const ffh_bulge_result *r) { return r ? r->hit_mm.data() : nullptr; }
const uint8_t *ffh_bulge_result_hit_bulge_type(const ffh_bulge_result *r) { return r ? r->hit_type.data() : nullptr; }
const uint8_t *ffh_bulge_result_hit_bulge_position(const ffh_bulge_result *r) { return r ? r->hit_pos.data() : nullptr; }
void ffh_bulge_result_free(ffh_bulge_result *r) { delete r; }

}  // extern "C"

// =====================================================================================================================
// multi-GPU reduction of the per-guide aggregates: pack / mask / unpack around the three collectives (dist.py)
// =====================================================================================================================
namespace ffh {

// prior of a shard = positions of the shards before it in database order, saturated like every running total (CRISPRSiteOT.scala:45)
__global__ void k_exchange_prior(const uint32_t *__restrict__ all_totals, uint32_t n, uint32_t rank, uint32_t clamp, uint32_t *__restrict__ prior) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint64_t sum = 0;
    for (uint32_t r = 0; r < rank; ++r) sum += all_totals[(size_t)r * n + g];
    prior[g] = (uint32_t)(sum < clamp ? sum : clamp);
}

// lanes of the MAX collective: overflow, cfd_max, jost_max, -closest (so that the MAX delivers the MIN); lanes of the SUM
// collective: n_hits, ot_count, hist[5], in_genome, n_scored, closest_count (filled by k_exchange_mask); f64 sums gathered apart
__global__ void k_exchange_pack(const GuideSummary *__restrict__ s, uint32_t n, double *__restrict__ mx, int32_t *__restrict__ sums, double *__restrict__ fsum) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const GuideSummary v = s[g];
    mx[4 * g + 0] = (double)v.overflow; mx[4 * g + 1] = v.cfd_max; mx[4 * g + 2] = v.jost_max; mx[4 * g + 3] = -(double)v.closest;
    int32_t *o = sums + 10 * (size_t)g;
    o[0] = (int32_t)v.n_hits; o[1] = (int32_t)v.ot_count;
    for (int k = 0; k < 5; ++k) o[2 + k] = (int32_t)v.hist[k];
    o[7] = (int32_t)v.in_genome; o[8] = (int32_t)v.n_scored; o[9] = 0;
    fsum[3 * (size_t)g + 0] = v.cfd_sum; fsum[3 * (size_t)g + 1] = v.hsu_sum; fsum[3 * (size_t)g + 2] = v.jost_sum;
}
// after the MAX collective: only the ranks that hold the globally closest level contribute their count (ClosestHit.scala:62-67)
__global__ void k_exchange_mask(const GuideSummary *__restrict__ s, uint32_t n, const double *__restrict__ mx, int32_t *__restrict__ sums) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    sums[10 * (size_t)g + 9] = ((double)s[g].closest == -mx[4 * g + 3]) ? (int32_t)s[g].closest_count : 0;
}
// fsum_all = [world][n][3]: added in rank order = database order of the shards (deterministic)
__global__ void k_exchange_unpack(GuideSummary *__restrict__ s, uint32_t n, const double *__restrict__ mx, const int32_t *__restrict__ sums,
                                  const double *__restrict__ fsum_all, uint32_t world) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    GuideSummary v;
    const int32_t *o = sums + 10 * (size_t)g;
    v.n_hits = (uint32_t)o[0]; v.ot_count = (uint32_t)o[1];
    for (int k = 0; k < 5; ++k) v.hist[k] = (uint32_t)o[2 + k];
    v.in_genome = (uint32_t)o[7]; v.n_scored = (uint32_t)o[8]; v.closest_count = (uint32_t)o[9];
    v.overflow = (uint32_t)mx[4 * g + 0]; v.cfd_max = mx[4 * g + 1]; v.jost_max = mx[4 * g + 2];
    v.closest = (uint32_t)(-mx[4 * g + 3]);
    double a = 0, b = 0, c = 0;
    for (uint32_t r = 0; r < world; ++r) {
        const double *f = fsum_all + ((size_t)r * n + g) * 3;
        if (r == 0) { a = f[0]; b = f[1]; c = f[2]; } else { a += f[0]; b += f[1]; c += f[2]; }
    }
    v.cfd_sum = a; v.hsu_sum = b; v.jost_sum = c;
    s[g] = v;
}

}  // namespace ffh

extern "C" {

int ffh_use_stream(ffh_ctx *ctx, void *hip_stream, int on) {
    if (!ctx) return FFH_E_ARG;
    FFH_HIP(hipSetDevice(ctx->device));
    FFH_HIP(hipStreamSynchronize(ctx->st));  // nothing of the old stream may still be in flight when the order changes
    ctx->borrowed = on != 0;
    ctx->st = ctx->borrowed ? (hipStream_t)hip_stream : ctx->own_st;
    return FFH_OK;
}

// the aggregates of this shard as if it were the first one (prior 0), and its saturated totals: one pass of the fused epilogue
static int shard_epilogue(ffh_ctx *ctx, int max_offtargets, unsigned flags, const uint32_t *d_prior, const uint32_t *d_fix_totals, void *d_summaries,
                          uint32_t *d_totals) {
    if (!ctx->scanned) { ctx->err = "ffh_scan has not run"; return FFH_E_STATE; }
    if (max_offtargets < 0) { ctx->err = "bad argument"; return FFH_E_ARG; }
    { const int rc = check_bound(ctx, max_offtargets); if (rc) return rc; }
    FFH_HIP(hipSetDevice(ctx->device));
    FFH_HIP(fence_in(ctx));
    const uint32_t G = ctx->n_guides;
    FFH_HIP(ctx->n_ret.reserve((size_t)G + 1));
    if (!d_fix_totals) FFH_HIP(hipEventRecord(ctx->ev[7], ctx->st));
    if (G) hipLaunchKernelGGL(k_guide_epilogue, dim3(blocks_for(G, 4)), dim3(256), 0, ctx->st, ctx->seg_begin.p, ctx->seg_end.p,
                              (const uint64_t *)(ctx->hit_t_ready ? ctx->hit_t.p : nullptr), (const uint64_t *)ctx->hits_sorted, (const uint64_t *)ctx->targets.p, ctx->tbits,
                              d_prior, ctx->guides.p, ctx->geo, ctx->d_tab, G, (uint32_t)max_offtargets, (flags & FFH_FINALIZE_JOST) ? 1 : 0, ctx->n_ret.p,
                              (GuideSummary *)d_summaries, d_totals, d_fix_totals, (GuideSummary *)nullptr);
    FFH_HIP(hipGetLastError());
    if (!d_fix_totals) { FFH_HIP(hipEventRecord(ctx->ev[1], ctx->st)); ctx->finalize_timing_pending = true; }
    FFH_HIP(fence_out(ctx));
    return FFH_OK;
}
int ffh_finalize_shard(ffh_ctx *ctx, int max_offtargets, unsigned flags, void *d_summaries, uint32_t *d_totals) {
    if (!ctx || !d_summaries || !d_totals) return FFH_E_ARG;
    return shard_epilogue(ctx, max_offtargets, flags, nullptr, nullptr, d_summaries, d_totals);
}
int ffh_finalize_shard_fixup(ffh_ctx *ctx, int max_offtargets, unsigned flags, const uint32_t *d_prior, const uint32_t *d_totals, void *d_summaries) {
    if (!ctx || !d_prior || !d_totals || !d_summaries) return FFH_E_ARG;
    return shard_epilogue(ctx, max_offtargets, flags, d_prior, d_totals, d_summaries, nullptr);
}
int ffh_exchange_prior(ffh_ctx *ctx, const uint32_t *d_all_totals, uint32_t n, uint32_t rank, uint32_t clamp, uint32_t *d_prior) {
    if (!ctx || !d_all_totals || !d_prior) return FFH_E_ARG;
    FFH_HIP(hipSetDevice(ctx->device));
    FFH_HIP(fence_in(ctx));
    if (n) hipLaunchKernelGGL(k_exchange_prior, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->st, d_all_totals, n, rank, clamp, d_prior);
    FFH_HIP(hipGetLastError());
    FFH_HIP(fence_out(ctx));
    return FFH_OK;
}

int ffh_exchange_pack(ffh_ctx *ctx, const void *d_summaries, uint32_t n, double *d_max, int32_t *d_sum, double *d_fsum) {
    if (!ctx || !d_summaries || !d_max || !d_sum || !d_fsum) return FFH_E_ARG;
    FFH_HIP(hipSetDevice(ctx->device));
    FFH_HIP(fence_in(ctx));
    if (n) hipLaunchKernelGGL(k_exchange_pack, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->st, (const GuideSummary *)d_summaries, n, d_max, d_sum, d_fsum);
    FFH_HIP(hipGetLastError());
    FFH_HIP(fence_out(ctx));
    return FFH_OK;
}
int ffh_exchange_mask(ffh_ctx *ctx, const void *d_summaries, uint32_t n, const double *d_max_reduced, int32_t *d_sum) {
    if (!ctx || !d_summaries || !d_max_reduced || !d_sum) return FFH_E_ARG;
    FFH_HIP(hipSetDevice(ctx->device));
    FFH_HIP(fence_in(ctx));  // the collective ran on another stream
    if (n) hipLaunchKernelGGL(k_exchange_mask, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->st, (const GuideSummary *)d_summaries, n, d_max_reduced, d_sum);
    FFH_HIP(hipGetLastError());
    FFH_HIP(fence_out(ctx));
    return FFH_OK;
}
int ffh_exchange_unpack(ffh_ctx *ctx, void *d_summaries, uint32_t n, const double *d_max_reduced, const int32_t *d_sum_reduced, const double *d_fsum_all,
                        uint32_t world) {
    if (!ctx || !d_summaries || !d_max_reduced || !d_sum_reduced || !d_fsum_all || !world) return FFH_E_ARG;
    FFH_HIP(hipSetDevice(ctx->device));
    FFH_HIP(fence_in(ctx));
    if (n) hipLaunchKernelGGL(k_exchange_unpack, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->st, (GuideSummary *)d_summaries, n, d_max_reduced, d_sum_reduced, d_fsum_all, world);
    FFH_HIP(hipGetLastError());
    FFH_HIP(fence_out(ctx));
    return FFH_OK;
}

}  // extern "C"

// =====================================================================================================================
// the bin-sharded discover with the collectives inside the library (RCCL; ffh_comm.hpp)
// =====================================================================================================================
#include "ffh_comm.hpp"
