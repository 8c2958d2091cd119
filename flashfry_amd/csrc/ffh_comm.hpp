// ffh_comm.hpp -- the bin-sharded discover INSIDE the library: scan of every shard + the exchange of SURVEY.md section 8e, with the
// collectives issued by the library itself on the contexts' streams (included by ffh_api.hip; needs ffh_ctx).
//
// Replaces, for N GPUs, what the reference's single traverser does alone (reference/traverser/Traverser.scala:38-61 chosen in
// modules/OffTargetDiscovery.scala:119-135): bins are sharded statically and contiguously (BinaryHeader.scala:54 balances them), every
// shard holds all guides, and per discover the shards exchange
//   (1) their per-guide position totals, saturated at maximumOffTargets  -> all-gather -> prior of every shard, so that the ordered
//       cut-off of CRISPRSiteOT.addOT / full (crispr/CRISPRSiteOT.scala:39-46) continues across shards in database order;
//   (2) the per-guide aggregates: MAX over (overflow, cfd_max, jost_max, -closest), SUM over the integer lanes (closest-hit count masked
//       to the winning level), all-gather of the three f64 sums which are then added in shard order (= database order, deterministic).
// Three transports, one code path above them:
//   RANK   one process per GPU (bench.py --gpus N, an MPI / torch.distributed host): ncclCommInitRank, the caller distributes the
//          128-byte unique id;
//   ALL    one process drives several GPUs (the CLI's --gpus N, a JVM that owns the node): ncclCommInitAll, grouped calls;
//   COPY   several shards on ONE device (tests and rehearsals of an N-way run on a single GPU; RCCL refuses duplicate devices):
//          all-gather = device-to-device copies ordered by events, all-reduce = all-gather + a local reduction kernel.
// RCCL (librccl.so.1, 570 MB) is opened with dlopen the first time a communicator needs it: a single-GPU discover never pays its
// load time (0.2 s warm, more cold) and the library stays loadable where RCCL is absent.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace ffh {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

static RcclApi *rccl_api() {   // nullptr-safe: check ->lib
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) { api.err = std::string("RCCL is not available: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
        bool ok = true;
        auto sym = [&](const char *n) { void *p = dlsym(api.lib, n); if (!p) { ok = false; api.err = std::string("RCCL lacks ") + n; } return p; };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(api.lib); api.lib = nullptr; }
    });
    return &api;
}

// COPY transport: out[i] = max / sum over the world's rows of all[world][len]
__global__ void k_comm_reduce_max_f64(const double *__restrict__ all, uint32_t world, uint64_t len, double *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double v = all[i];
    for (uint32_t r = 1; r < world; ++r) v = fmax(v, all[(uint64_t)r * len + i]);
    out[i] = v;
}
__global__ void k_comm_reduce_sum_i32(const int32_t *__restrict__ all, uint32_t world, uint64_t len, int32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    int32_t v = all[i];
    for (uint32_t r = 1; r < world; ++r) v += all[(uint64_t)r * len + i];
    out[i] = v;
}

}  // namespace ffh

enum { FFH_COMM_COPY = 0, FFH_COMM_ALL = 1, FFH_COMM_RANK = 2 };

struct ffh_comm {
    int mode = FFH_COMM_COPY;
    int world = 1, first = 0;            // shards in total; number of this process's first shard
    std::vector<ffh_ctx *> ctx;          // this process's shards, in database order
    std::vector<ncclComm_t> nccl;        // one per local shard (ALL) or one (RANK)
    struct Buf {
        DevBuf<uint32_t> totals, all_totals, prior;
        DevBuf<GuideSummary> summ;
        DevBuf<double> mx, mx_all, fsum, fsum_all;
        DevBuf<int32_t> sums, sums_all;
        hipEvent_t ev = nullptr;
    };
    std::vector<std::unique_ptr<Buf>> buf;
    std::string err;
    std::mutex err_m;                    // ffh_comm_shard_lists may be called for different local shards from different host threads
    uint32_t n_guides = 0;
    int max_ot = 0;
    bool exchanged = false;
    double scan_ms = 0, exchange_ms = 0;  // host wall time of the last ffh_discover_sharded: scans (all local shards), exchange + copy-out
};

namespace {

#define FFC_HIP(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) { cm->err = std::string(#expr) + ": " + hipGetErrorString(e_); return FFH_E_HIP; } \
    } while (0)
#define FFC_NCCL(expr)                                                                                \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess) { cm->err = std::string(#expr) + ": " + rccl_api()->GetErrorString(r_); return FFH_E_HIP; } \
    } while (0)

// all-gather of `len` elements of T per shard: src(i) -> dst(i)[world][len] on every local shard; stream-ordered on the contexts' streams
template <typename T, typename Src, typename Dst>
int comm_all_gather(ffh_comm *cm, uint64_t len, ncclDataType_t dt, Src src, Dst dst) {
    const size_t L = cm->ctx.size();
    if (cm->mode == FFH_COMM_COPY) {
        for (size_t i = 0; i < L; ++i) { FFC_HIP(hipSetDevice(cm->ctx[i]->device)); FFC_HIP(hipEventRecord(cm->buf[i]->ev, cm->ctx[i]->st)); }
        for (size_t j = 0; j < L; ++j) {
            FFC_HIP(hipSetDevice(cm->ctx[j]->device));
            for (size_t i = 0; i < L; ++i) {
                if (i != j) FFC_HIP(hipStreamWaitEvent(cm->ctx[j]->st, cm->buf[i]->ev, 0));
                FFC_HIP(hipMemcpyAsync(dst(j) + (uint64_t)i * len, src(i), len * sizeof(T), hipMemcpyDeviceToDevice, cm->ctx[j]->st));
            }
        }
        return FFH_OK;
    }
    RcclApi *R = rccl_api();
    if (L > 1) FFC_NCCL(R->GroupStart());
    for (size_t i = 0; i < L; ++i) {
        FFC_HIP(hipSetDevice(cm->ctx[i]->device));
        FFC_NCCL(R->AllGather(src(i), dst(i), len, dt, cm->nccl[i], cm->ctx[i]->st));
    }
    if (L > 1) FFC_NCCL(R->GroupEnd());
    return FFH_OK;
}

}  // namespace

extern "C" {

int ffh_comm_unique_id(void *id128) {
    if (!id128) return FFH_E_ARG;
    RcclApi *R = rccl_api();
    if (!R->lib) { g_create_error = R->err; return FFH_E_STATE; }
    static_assert(sizeof(ncclUniqueId) == 128, "the C ABI hands the id over as 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = R->GetUniqueId(&id);
    if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + R->GetErrorString(r); return FFH_E_HIP; }
    std::memcpy(id128, &id, 128);
    return FFH_OK;
}

static ffh_comm *comm_new(ffh_ctx *const *ctxs, int n) {
    ffh_comm *cm = new (std::nothrow) ffh_comm();
    if (!cm) { g_create_error = "out of memory"; return nullptr; }
    cm->ctx.assign(ctxs, ctxs + n);
    for (int i = 0; i < n; ++i) {
        cm->buf.emplace_back(new ffh_comm::Buf());
        if (hipSetDevice(ctxs[i]->device) != hipSuccess || hipEventCreateWithFlags(&cm->buf.back()->ev, hipEventDisableTiming) != hipSuccess) {
            g_create_error = "HIP initialisation of the communicator failed";
            ffh_comm_destroy(cm);
            return nullptr;
        }
    }
    return cm;
}

void ffh_comm_destroy(ffh_comm *cm) {
    if (!cm) return;
    for (size_t i = 0; i < cm->ctx.size(); ++i) {
        (void)hipSetDevice(cm->ctx[i]->device);
        (void)hipStreamSynchronize(cm->ctx[i]->st);
        if (i < cm->buf.size() && cm->buf[i]->ev) (void)hipEventDestroy(cm->buf[i]->ev);
        if (i < cm->buf.size()) cm->buf[i].reset();   // (device buffers are freed on their device)
    }
    for (auto c : cm->nccl) if (c) (void)rccl_api()->CommDestroy(c);
    delete cm;
}

int ffh_comm_create_local(ffh_ctx *const *ctxs, int n, ffh_comm **out) {
    if (!ctxs || n < 1 || !out) return FFH_E_ARG;
    for (int i = 0; i < n; ++i) if (!ctxs[i]) return FFH_E_ARG;
    ffh_comm *cm = comm_new(ctxs, n);
    if (!cm) return FFH_E_NOMEM;
    cm->world = n; cm->first = 0;
    bool distinct = n > 1;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) if (ctxs[i]->device == ctxs[j]->device) distinct = false;
    const char *force = std::getenv("FFH_COMM");   // "copy": never RCCL (A/B, boxes without it)
    if (distinct && !(force && std::strcmp(force, "copy") == 0)) {
        RcclApi *R = rccl_api();
        if (!R->lib) { g_create_error = R->err; ffh_comm_destroy(cm); return FFH_E_STATE; }
        std::vector<int> devs(n);
        for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
        cm->nccl.assign(n, nullptr);
        const ncclResult_t r = R->CommInitAll(cm->nccl.data(), n, devs.data());
        if (r != ncclSuccess) { g_create_error = std::string("ncclCommInitAll: ") + R->GetErrorString(r); cm->nccl.clear(); ffh_comm_destroy(cm); return FFH_E_HIP; }
        cm->mode = FFH_COMM_ALL;
    } else cm->mode = FFH_COMM_COPY;
    *out = cm;
    return FFH_OK;
}

int ffh_comm_create_rank(ffh_ctx *ctx, int rank, int world, const void *id128, ffh_comm **out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || !id128) return FFH_E_ARG;
    RcclApi *R = rccl_api();
    if (!R->lib) { g_create_error = R->err; return FFH_E_STATE; }
    ffh_comm *cm = comm_new(&ctx, 1);
    if (!cm) return FFH_E_NOMEM;
    cm->world = world; cm->first = rank; cm->mode = FFH_COMM_RANK;
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    cm->nccl.assign(1, nullptr);
    (void)hipSetDevice(ctx->device);
    const ncclResult_t r = R->CommInitRank(&cm->nccl[0], world, id, rank);
    if (r != ncclSuccess) { g_create_error = std::string("ncclCommInitRank: ") + R->GetErrorString(r); cm->nccl.clear(); ffh_comm_destroy(cm); return FFH_E_HIP; }
    *out = cm;
    return FFH_OK;
}

const char *ffh_comm_last_error(const ffh_comm *cm) { return cm ? cm->err.c_str() : g_create_error.c_str(); }
int ffh_comm_world(const ffh_comm *cm) { return cm ? cm->world : 0; }
int ffh_comm_first_shard(const ffh_comm *cm) { return cm ? cm->first : 0; }
int ffh_comm_local_shards(const ffh_comm *cm) { return cm ? (int)cm->ctx.size() : 0; }
int ffh_comm_transport(const ffh_comm *cm) { return cm ? cm->mode : -1; }
int ffh_comm_timings(const ffh_comm *cm, double *scan_ms, double *exchange_ms) {
    if (!cm) return FFH_E_ARG;
    if (scan_ms) *scan_ms = cm->scan_ms;
    if (exchange_ms) *exchange_ms = cm->exchange_ms;
    return FFH_OK;
}

// the exchange of the shards' aggregates after every local shard has been scanned: stream-ordered on the contexts' streams, no host
// round trip until the reduced summaries are copied out
static int comm_exchange(ffh_comm *cm, uint32_t G, int max_ot, unsigned flags, ffh_guide_summary *summaries_out) {
    const size_t L = cm->ctx.size();
    const uint32_t W = (uint32_t)cm->world;
    const bool copy = cm->mode == FFH_COMM_COPY;
    const unsigned jost = flags & FFH_FINALIZE_JOST;
    for (size_t i = 0; i < L; ++i) {
        ffh_ctx *ctx = cm->ctx[i];
        ffh_comm::Buf &b = *cm->buf[i];
        FFC_HIP(hipSetDevice(ctx->device));
        FFC_HIP(b.totals.reserve((size_t)G + 1)); FFC_HIP(b.all_totals.reserve((size_t)W * G + 1)); FFC_HIP(b.prior.reserve((size_t)G + 1));
        FFC_HIP(b.summ.reserve((size_t)G + 1));
        FFC_HIP(b.mx.reserve((size_t)4 * G + 1)); FFC_HIP(b.sums.reserve((size_t)10 * G + 1)); FFC_HIP(b.fsum.reserve((size_t)3 * G + 1));
        FFC_HIP(b.fsum_all.reserve((size_t)W * 3 * G + 1));
        if (copy) { FFC_HIP(b.mx_all.reserve((size_t)W * 4 * G + 1)); FFC_HIP(b.sums_all.reserve((size_t)W * 10 * G + 1)); }
        FFC_HIP(ctx->n_ret.reserve((size_t)G + 1));
        if (!ctx->scanned || ctx->n_guides != G) { cm->err = "a shard has not been scanned with this guide set"; return FFH_E_STATE; }
    }
    if (!G) return FFH_OK;
    auto epilogue = [&](size_t i, const uint32_t *d_prior, const uint32_t *d_fix, uint32_t *d_totals) {
        ffh_ctx *ctx = cm->ctx[i];
        (void)hipSetDevice(ctx->device);
        hipLaunchKernelGGL(k_guide_epilogue, dim3(blocks_for(G, 4)), dim3(256), 0, ctx->st, ctx->seg_begin.p, ctx->seg_end.p,
                           (const uint64_t *)(ctx->hit_t_ready ? ctx->hit_t.p : nullptr), (const uint64_t *)ctx->hits_sorted, (const uint64_t *)ctx->targets.p, ctx->tbits,
                           d_prior, ctx->guides.p, ctx->geo, ctx->d_tab, G, (uint32_t)max_ot, jost ? 1 : 0, ctx->n_ret.p, cm->buf[i]->summ.p, d_totals, d_fix,
                           (GuideSummary *)nullptr);
    };
    // every shard aggregates as if it were the first one; the same pass yields its saturated totals
    for (size_t i = 0; i < L; ++i) epilogue(i, nullptr, nullptr, cm->buf[i]->totals.p);
    int rc = comm_all_gather<uint32_t>(cm, G, ncclUint32, [&](size_t i) { return (const uint32_t *)cm->buf[i]->totals.p; }, [&](size_t j) { return cm->buf[j]->all_totals.p; });
    if (rc) return rc;
    for (size_t i = 0; i < L; ++i) {
        ffh_ctx *ctx = cm->ctx[i];
        ffh_comm::Buf &b = *cm->buf[i];
        FFC_HIP(hipSetDevice(ctx->device));
        hipLaunchKernelGGL(k_exchange_prior, dim3(blocks_for(G, 256)), dim3(256), 0, ctx->st, (const uint32_t *)b.all_totals.p, G, (uint32_t)(cm->first + (int)i), (uint32_t)max_ot, b.prior.p);
        // ... and only the guides whose cut-off the shards before it move are aggregated again
        epilogue(i, b.prior.p, b.totals.p, nullptr);
        hipLaunchKernelGGL(k_exchange_pack, dim3(blocks_for(G, 256)), dim3(256), 0, ctx->st, (const GuideSummary *)b.summ.p, G, b.mx.p, b.sums.p, b.fsum.p);
    }
    RcclApi *R = copy ? nullptr : rccl_api();
    // MAX over (overflow, cfd_max, jost_max, -closest)
    if (copy) {
        rc = comm_all_gather<double>(cm, (uint64_t)4 * G, ncclFloat64, [&](size_t i) { return (const double *)cm->buf[i]->mx.p; }, [&](size_t j) { return cm->buf[j]->mx_all.p; });
        if (rc) return rc;
        for (size_t i = 0; i < L; ++i) {
            FFC_HIP(hipSetDevice(cm->ctx[i]->device));
            hipLaunchKernelGGL(k_comm_reduce_max_f64, dim3(blocks_for((uint64_t)4 * G, 256)), dim3(256), 0, cm->ctx[i]->st, (const double *)cm->buf[i]->mx_all.p, W, (uint64_t)4 * G, cm->buf[i]->mx.p);
        }
    } else {
        if (L > 1) FFC_NCCL(R->GroupStart());
        for (size_t i = 0; i < L; ++i) { FFC_HIP(hipSetDevice(cm->ctx[i]->device)); FFC_NCCL(R->AllReduce(cm->buf[i]->mx.p, cm->buf[i]->mx.p, (size_t)4 * G, ncclFloat64, ncclMax, cm->nccl[i], cm->ctx[i]->st)); }
        if (L > 1) FFC_NCCL(R->GroupEnd());
    }
    for (size_t i = 0; i < L; ++i) {
        FFC_HIP(hipSetDevice(cm->ctx[i]->device));
        hipLaunchKernelGGL(k_exchange_mask, dim3(blocks_for(G, 256)), dim3(256), 0, cm->ctx[i]->st, (const GuideSummary *)cm->buf[i]->summ.p, G, (const double *)cm->buf[i]->mx.p, cm->buf[i]->sums.p);
    }
    // SUM over the integer lanes
    if (copy) {
        rc = comm_all_gather<int32_t>(cm, (uint64_t)10 * G, ncclInt32, [&](size_t i) { return (const int32_t *)cm->buf[i]->sums.p; }, [&](size_t j) { return cm->buf[j]->sums_all.p; });
        if (rc) return rc;
        for (size_t i = 0; i < L; ++i) {
            FFC_HIP(hipSetDevice(cm->ctx[i]->device));
            hipLaunchKernelGGL(k_comm_reduce_sum_i32, dim3(blocks_for((uint64_t)10 * G, 256)), dim3(256), 0, cm->ctx[i]->st, (const int32_t *)cm->buf[i]->sums_all.p, W, (uint64_t)10 * G, cm->buf[i]->sums.p);
        }
    } else {
        if (L > 1) FFC_NCCL(R->GroupStart());
        for (size_t i = 0; i < L; ++i) { FFC_HIP(hipSetDevice(cm->ctx[i]->device)); FFC_NCCL(R->AllReduce(cm->buf[i]->sums.p, cm->buf[i]->sums.p, (size_t)10 * G, ncclInt32, ncclSum, cm->nccl[i], cm->ctx[i]->st)); }
        if (L > 1) FFC_NCCL(R->GroupEnd());
    }
    // the f64 sums: gathered, then added in shard order
    rc = comm_all_gather<double>(cm, (uint64_t)3 * G, ncclFloat64, [&](size_t i) { return (const double *)cm->buf[i]->fsum.p; }, [&](size_t j) { return cm->buf[j]->fsum_all.p; });
    if (rc) return rc;
    for (size_t i = 0; i < L; ++i) {
        FFC_HIP(hipSetDevice(cm->ctx[i]->device));
        hipLaunchKernelGGL(k_exchange_unpack, dim3(blocks_for(G, 256)), dim3(256), 0, cm->ctx[i]->st, cm->buf[i]->summ.p, G, (const double *)cm->buf[i]->mx.p, (const int32_t *)cm->buf[i]->sums.p,
                           (const double *)cm->buf[i]->fsum_all.p, W);
    }
    FFC_HIP(hipGetLastError());
    if (summaries_out) {
        FFC_HIP(hipSetDevice(cm->ctx[0]->device));
        FFC_HIP(hipMemcpyAsync(summaries_out, cm->buf[0]->summ.p, (size_t)G * sizeof(ffh_guide_summary), hipMemcpyDeviceToHost, cm->ctx[0]->st));
    }
    for (size_t i = 0; i < L; ++i) { FFC_HIP(hipSetDevice(cm->ctx[i]->device)); FFC_HIP(hipStreamSynchronize(cm->ctx[i]->st)); }
    return FFH_OK;
}

int ffh_discover_sharded(ffh_comm *cm, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets, unsigned flags, ffh_guide_summary *summaries_out) {
    if (!cm || (n_guides && !guides) || max_mismatch < 0 || max_offtargets < 0) { if (cm) cm->err = "bad argument"; return FFH_E_ARG; }
    const size_t L = cm->ctx.size();
    cm->exchanged = false;
    const auto t0 = std::chrono::steady_clock::now();
    // the scans: every local shard on its own host thread (a scan reads its hit count back before it orders the hits, so one thread
    // would run the GPUs one after the other); each shard is bounded by its own totals (the prior of the lower shards could only
    // retire more guides)
    std::vector<int> rcs(L, FFH_OK);
    auto scan = [&](size_t i) { rcs[i] = ffh_scan_bounded(cm->ctx[i], guides, n_guides, max_mismatch, max_offtargets); };
    if (L == 1) scan(0);
    else {
        std::vector<std::thread> th;
        for (size_t i = 1; i < L; ++i) th.emplace_back(scan, i);
        scan(0);
        for (auto &t : th) t.join();
    }
    for (size_t i = 0; i < L; ++i)
        if (rcs[i]) { cm->err = "shard " + std::to_string(cm->first + (int)i) + ": " + cm->ctx[i]->err; return rcs[i]; }
    const auto t1 = std::chrono::steady_clock::now();
    const int rc = comm_exchange(cm, n_guides, max_offtargets, flags, summaries_out);
    if (rc) return rc;
    cm->n_guides = n_guides; cm->max_ot = max_offtargets; cm->exchanged = true;
    cm->scan_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    cm->exchange_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    return FFH_OK;
}

// the exchange alone, for callers that scanned the shards themselves (ffh_scan / ffh_scan_bounded on every local context)
int ffh_comm_exchange(ffh_comm *cm, uint32_t n_guides, int max_offtargets, unsigned flags, ffh_guide_summary *summaries_out) {
    if (!cm || max_offtargets < 0) { if (cm) cm->err = "bad argument"; return FFH_E_ARG; }
    cm->exchanged = false;
    const auto t1 = std::chrono::steady_clock::now();
    const int rc = comm_exchange(cm, n_guides, max_offtargets, flags, summaries_out);
    if (rc) return rc;
    cm->n_guides = n_guides; cm->max_ot = max_offtargets; cm->exchanged = true;
    cm->exchange_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    return FFH_OK;
}

int ffh_comm_shard_lists(ffh_comm *cm, int local_shard, unsigned flags, ffh_result **out) {
    if (!cm || !out || local_shard < 0 || (size_t)local_shard >= cm->ctx.size()) { if (cm) cm->err = "bad argument"; return FFH_E_ARG; }
    if (!cm->exchanged) { cm->err = "ffh_discover_sharded has not run"; return FFH_E_STATE; }
    ffh_ctx *ctx = cm->ctx[(size_t)local_shard];
    const int rc = ffh_finalize(ctx, cm->buf[(size_t)local_shard]->prior.p, cm->max_ot, (flags & ~FFH_FINALIZE_SUMMARIES_ONLY) | FFH_FINALIZE_PRIOR_ON_DEVICE, out);
    if (rc) { std::lock_guard<std::mutex> g(cm->err_m); cm->err = ctx->err; }
    return rc;
}

int ffh_comm_device_summaries(ffh_comm *cm, int local_shard, const void **device_summaries) {
    if (!cm || !device_summaries || local_shard < 0 || (size_t)local_shard >= cm->ctx.size()) return FFH_E_ARG;
    if (!cm->exchanged) { cm->err = "ffh_discover_sharded has not run"; return FFH_E_STATE; }
    *device_summaries = cm->buf[(size_t)local_shard]->summ.p;
    return FFH_OK;
}

}  // extern "C"
