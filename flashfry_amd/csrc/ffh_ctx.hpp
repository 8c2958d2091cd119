// ffh_ctx.hpp -- the context and what it owns: device buffers, the page-locked result pool, ffh_result, ffh_ctx, the polled host wait
// Part of the ONE translation unit ffh_api.hip (included there, in order; not a header of its own).
using namespace ffh;

static thread_local std::string g_create_error;  // what ffh_last_error(NULL) returns: per thread, contexts are created from several threads
namespace ffh {
void set_global_error(const std::string &m) { g_create_error = m; }
}  // namespace ffh

// (on failure the stream is waited for before the function returns: an asynchronous copy issued a line earlier into one of the function's locals
// must not land in a dead stack frame -- profiles/r06/uaf_analysis.md, appendix.  Inside a stream capture the wait is refused by the runtime
// and costs nothing.)
#define FFH_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                               \
            if (ctx->st) { (void)hipStreamSynchronize(ctx->st); (void)hipGetLastError(); }              \
            return FFH_E_HIP;                                                                           \
        }                                                                                               \
    } while (0)

#include "ffh_abi_guard.hpp"   // FFH_CATCH: no C++ exception crosses the C ABI
#include "ffh_devbuf.hpp"   // DevBuf<T>: the device allocation every buffer of a context is (owned, or an alias of another context's)
namespace {

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
    void alias(const Image &o) {   // the same image through another context (ffh_ctx_share_db): nothing is copied, nothing is owned
        width = o.width; direct = o.direct; rest = o.rest; live_bits = o.live_bits;
        bstart.alias(o.bstart); gstart.alias(o.gstart); gwords.alias(o.gwords); tidx.alias(o.tidx); live.alias(o.live);
    }
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
        // (called from ~ffh_result: must not throw -- a block that cannot be listed is freed instead of pooled)
        try { free_blocks.emplace_back(p, cap); } catch (...) { (void)hipHostFree(p); }
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
    ffh_ctx *db_owner = nullptr;   // ffh_ctx_share_db: this context scans db_owner's resident database (targets, positions, the two scan images are aliases)
    std::atomic<int> db_sharers{0};   // ... and so many contexts scan this one's: it must not load, rebuild or drop what they alias
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
    int pending_setup = 0;   // scan_impl -> prepare_side: 1 = the next k_guide_keys launch clears the compare launch's per-launch counters, 2 = and the per-scan ones
    struct SideScratch { DevBuf<uint32_t> part_fill, part_hist, gp_start, by_part; } side_scr[2];
    DevBuf<uint32_t> tmp_keys, tmp_tidx;                    // build_image's temporaries
    DevBuf<uint32_t> wl_count[2];                             // work entries per batch of buckets + per block of 1024 batches
    DevBuf<WorkEntry> wl_list[2];                             // the compare kernel's work list, per image
    DevBuf<uint64_t> scan_tmp64;
    DevBuf<uint32_t> sort_table, sort_offs, heavy_list;
    DevBuf<unsigned long long> sort_status;   // look-back words of the one-sweep radix passes (ffh_prims.hpp)
    std::map<std::pair<int, int>, std::vector<uint32_t>> pattern_cache;

    // finalize scratch
    DevBuf<uint32_t> n_ret, ot_count, full, prior, out_cnt, out_tidx, totals, hit_pre;
    DevBuf<unsigned long long> sub_hist, kept_ctr;   // ... and what stops a guide inside a slab (ffh_kernels.hpp: k_slab_subhist)
    DevBuf<uint16_t> hit_cnt;
    DevBuf<uint32_t> g_allow;
    DevBuf<uint8_t> g_thr;
    DevBuf<unsigned long long> totals64;   // a bounded scan: the positions the slab just scanned adds to every guide (k_slab_totals)
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
