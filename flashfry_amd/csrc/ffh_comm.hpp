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
    decltype(&ncclSend) Send = nullptr;     // (the slice exchange only: a runtime without them keeps the all-gather form)
    decltype(&ncclRecv) Recv = nullptr;
    std::string err;
};

static RcclApi *rccl_api() {   // nullptr-safe: check ->lib
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // FFH_RCCL_LIBRARY: another file name for the RCCL runtime (a site's own build; the tests name a missing one to take the
        // "no RCCL on this box" path)
        const char *forced = std::getenv("FFH_RCCL_LIBRARY");
        std::string why;
        for (const char *name : {forced ? forced : "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
            const char *e = dlerror();   // (one call: glibc hands the message out once and clears it)
            if (why.empty()) why = e ? e : "dlopen failed";
            if (forced) break;
        }
        if (!api.lib) { api.err = "RCCL is not available: " + why; return; }
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
        if (ok) { api.Send = (decltype(api.Send))dlsym(api.lib, "ncclSend"); api.Recv = (decltype(api.Recv))dlsym(api.lib, "ncclRecv"); }
        if (!ok) { dlclose(api.lib); api.lib = nullptr; }
    });
    return &api;
}

}  // namespace ffh
#include "ffh_exchange_kernels.hpp"   // k_exchange_reduce + the kernels of the exchange by guide slices (plain enough to be run on the CPU by tests/exchange_emul_main.cpp)

enum { FFH_COMM_COPY = 0, FFH_COMM_ALL = 1, FFH_COMM_RANK = 2 };
enum { FFH_EXCHANGE_GATHER = 0, FFH_EXCHANGE_SLICE = 1 };
static const char *const kSplitMessage = "the guide set was split (more raw hits than one scan holds): hit lists and device summaries are per call -- pass fewer guides per ffh_discover_sharded";
constexpr int kSplitGuides = -1000;   // internal: comm_exchange -> discover_sharded_split (never returned through the C ABI)

struct ffh_comm {
    int mode = FFH_COMM_COPY;
    int world = 1, first = 0;            // shards in total; number of this process's first shard
    std::vector<ffh_ctx *> ctx;          // this process's shards, in database order
    std::vector<ncclComm_t> nccl;        // one per local shard (ALL) or one (RANK)
    struct Buf {
        DevBuf<uint32_t> totals, prior, flag;    // the shard's own saturated totals, its prior (k_exchange_reduce), {crossing guides | failure bit}
        DevBuf<GuideSummary> summ, all, red;     // [G + 1] this shard's aggregates + status record; [world][G + 1] everybody's; [G] the reduced ones
        DevBuf<GuideSummary> send, red_slice;    // slice exchange: [world][sl + 1] packed by destination; [sl] the folded slice (red then is [world * sl])
        DevBuf<uint32_t> prior_all, prior_in;    // slice exchange: [world][sl + 1] priors of every shard for my slice (+ flag word); what came back
        hipEvent_t ev = nullptr, ev_done = nullptr;   // COPY transport: "my record is ready", "I have read everybody's"
    };
    std::vector<std::unique_ptr<Buf>> buf;
    std::string err;
    std::mutex err_m;                    // ffh_comm_shard_lists may be called for different local shards from different host threads
    uint32_t n_guides = 0;
    int max_ot = 0;
    bool exchanged = false;
    bool was_split = false;              // the last ffh_discover_sharded halved its guide set: hit lists / device summaries are refused
    uint32_t crossing = 0;               // guides of the last exchange whose cut-off fell inside a shard with a non-zero prior (second round)
    int exchange = FFH_EXCHANGE_GATHER;   // which form of the exchange (ffh_comm_set_exchange; FFH_EXCHANGE=slice at creation)
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
            FFC_HIP(hipEventRecord(cm->buf[j]->ev_done, cm->ctx[j]->st));
        }
        // nobody goes on -- and possibly rewrites its record -- before every peer has read it (a stream that ran ahead
        // reduced in place into the buffer a late reader was still copying from)
        for (size_t i = 0; i < L; ++i) {
            FFC_HIP(hipSetDevice(cm->ctx[i]->device));
            for (size_t j = 0; j < L; ++j) if (i != j) FFC_HIP(hipStreamWaitEvent(cm->ctx[i]->st, cm->buf[j]->ev_done, 0));
        }
        return FFH_OK;
    }
    RcclApi *R = rccl_api();
    if (L > 1) FFC_NCCL(R->GroupStart());
    int rc = FFH_OK;
    for (size_t i = 0; i < L && rc == FFH_OK; ++i) {   // (an error inside the group still closes it)
        if (hipSetDevice(cm->ctx[i]->device) != hipSuccess) { cm->err = "hipSetDevice failed inside the collective"; rc = FFH_E_HIP; break; }
        const ncclResult_t r = R->AllGather(src(i), dst(i), len, dt, cm->nccl[i], cm->ctx[i]->st);
        if (r != ncclSuccess) { cm->err = std::string("ncclAllGather: ") + R->GetErrorString(r); rc = FFH_E_HIP; }
    }
    if (L > 1) { const ncclResult_t r = R->GroupEnd(); if (r != ncclSuccess && rc == FFH_OK) { cm->err = std::string("ncclGroupEnd: ") + R->GetErrorString(r); rc = FFH_E_HIP; } }
    return rc;
}


// all-to-all of `len` elements of T per pair: shard i's src(i)[j * len ..] -> shard j's dst(j)[i * len ..]; stream-ordered on the contexts' streams
template <typename T, typename Src, typename Dst>
int comm_all_to_all(ffh_comm *cm, uint64_t len, ncclDataType_t dt, Src src, Dst dst) {
    const size_t L = cm->ctx.size();
    if (cm->mode == FFH_COMM_COPY) {
        for (size_t i = 0; i < L; ++i) { FFC_HIP(hipSetDevice(cm->ctx[i]->device)); FFC_HIP(hipEventRecord(cm->buf[i]->ev, cm->ctx[i]->st)); }
        for (size_t j = 0; j < L; ++j) {
            FFC_HIP(hipSetDevice(cm->ctx[j]->device));
            for (size_t i = 0; i < L; ++i) {
                if (i != j) FFC_HIP(hipStreamWaitEvent(cm->ctx[j]->st, cm->buf[i]->ev, 0));
                FFC_HIP(hipMemcpyAsync(dst(j) + (uint64_t)i * len, src(i) + (uint64_t)j * len, len * sizeof(T), hipMemcpyDeviceToDevice, cm->ctx[j]->st));
            }
            FFC_HIP(hipEventRecord(cm->buf[j]->ev_done, cm->ctx[j]->st));
        }
        for (size_t i = 0; i < L; ++i) {   // (nobody rewrites what it sent before every peer has read it: as in comm_all_gather)
            FFC_HIP(hipSetDevice(cm->ctx[i]->device));
            for (size_t j = 0; j < L; ++j) if (i != j) FFC_HIP(hipStreamWaitEvent(cm->ctx[i]->st, cm->buf[j]->ev_done, 0));
        }
        return FFH_OK;
    }
    RcclApi *R = rccl_api();
    if (!R->Send || !R->Recv) { cm->err = "this RCCL has no ncclSend / ncclRecv: the slice exchange is not available"; return FFH_E_STATE; }
    const int W = cm->world;
    FFC_NCCL(R->GroupStart());
    int rc = FFH_OK;
    for (size_t i = 0; i < L && rc == FFH_OK; ++i) {   // (an error inside the group still closes it)
        if (hipSetDevice(cm->ctx[i]->device) != hipSuccess) { cm->err = "hipSetDevice failed inside the collective"; rc = FFH_E_HIP; break; }
        for (int peer = 0; peer < W && rc == FFH_OK; ++peer) {
            ncclResult_t r = R->Send(src(i) + (uint64_t)peer * len, len, dt, peer, cm->nccl[i], cm->ctx[i]->st);
            if (r == ncclSuccess) r = R->Recv(dst(i) + (uint64_t)peer * len, len, dt, peer, cm->nccl[i], cm->ctx[i]->st);
            if (r != ncclSuccess) { cm->err = std::string("ncclSend / ncclRecv: ") + R->GetErrorString(r); rc = FFH_E_HIP; }
        }
    }
    { const ncclResult_t r = R->GroupEnd(); if (r != ncclSuccess && rc == FFH_OK) { cm->err = std::string("ncclGroupEnd: ") + R->GetErrorString(r); rc = FFH_E_HIP; } }
    return rc;
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
    try {
    cm->ctx.assign(ctxs, ctxs + n);
    { const char *e = std::getenv("FFH_EXCHANGE"); if (e && std::strcmp(e, "slice") == 0) cm->exchange = FFH_EXCHANGE_SLICE; }
    for (int i = 0; i < n; ++i) {
        cm->buf.emplace_back(new ffh_comm::Buf());
        if (hipSetDevice(ctxs[i]->device) != hipSuccess || hipEventCreateWithFlags(&cm->buf.back()->ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&cm->buf.back()->ev_done, hipEventDisableTiming) != hipSuccess) {
            g_create_error = "HIP initialisation of the communicator failed";
            ffh_comm_destroy(cm);
            return nullptr;
        }
    }
    } catch (...) { ffh_comm_destroy(cm); ffh_note_exception(&g_create_error, "out of host memory creating a communicator"); return nullptr; }
    return cm;
}

void ffh_comm_destroy(ffh_comm *cm) {
    if (!cm) return;
    for (size_t i = 0; i < cm->ctx.size(); ++i) {
        (void)hipSetDevice(cm->ctx[i]->device);
        (void)hipStreamSynchronize(cm->ctx[i]->st);
        if (i < cm->buf.size() && cm->buf[i]->ev) (void)hipEventDestroy(cm->buf[i]->ev);
        if (i < cm->buf.size() && cm->buf[i]->ev_done) (void)hipEventDestroy(cm->buf[i]->ev_done);
        if (i < cm->buf.size()) cm->buf[i].reset();   // (device buffers are freed on their device)
    }
    for (auto c : cm->nccl) if (c) (void)rccl_api()->CommDestroy(c);
    delete cm;
}

int ffh_comm_create_local(ffh_ctx *const *ctxs, int n, ffh_comm **out) try {
    if (!ctxs || n < 1 || !out) return FFH_E_ARG;
    for (int i = 0; i < n; ++i) if (!ctxs[i]) return FFH_E_ARG;
    ffh_comm *cm = comm_new(ctxs, n);
    if (!cm) return FFH_E_NOMEM;
    try {
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
    } catch (...) { ffh_comm_destroy(cm); throw; }
    *out = cm;
    return FFH_OK;
} FFH_CATCH(&g_create_error)

int ffh_comm_create_rank(ffh_ctx *ctx, int rank, int world, const void *id128, ffh_comm **out) try {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || !id128) return FFH_E_ARG;
    RcclApi *R = rccl_api();
    if (!R->lib) { g_create_error = R->err; return FFH_E_STATE; }
    ffh_comm *cm = comm_new(&ctx, 1);
    if (!cm) return FFH_E_NOMEM;
    try {
    cm->world = world; cm->first = rank; cm->mode = FFH_COMM_RANK;
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    cm->nccl.assign(1, nullptr);
    (void)hipSetDevice(ctx->device);
    const ncclResult_t r = R->CommInitRank(&cm->nccl[0], world, id, rank);
    if (r != ncclSuccess) { g_create_error = std::string("ncclCommInitRank: ") + R->GetErrorString(r); cm->nccl.clear(); ffh_comm_destroy(cm); return FFH_E_HIP; }
    } catch (...) { ffh_comm_destroy(cm); throw; }
    *out = cm;
    return FFH_OK;
} FFH_CATCH(&g_create_error)

void *ffh_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // (portable: whichever device the caller's contexts sit on)
    return p;
}
void ffh_host_free(void *p) { if (p) (void)hipHostFree(p); }

const char *ffh_comm_last_error(const ffh_comm *cm) { return cm ? cm->err.c_str() : g_create_error.c_str(); }
int ffh_comm_world(const ffh_comm *cm) { return cm ? cm->world : 0; }
int ffh_comm_first_shard(const ffh_comm *cm) { return cm ? cm->first : 0; }
int ffh_comm_local_shards(const ffh_comm *cm) { return cm ? (int)cm->ctx.size() : 0; }
int ffh_comm_transport(const ffh_comm *cm) { return cm ? cm->mode : -1; }
int ffh_comm_set_exchange(ffh_comm *cm, int mode) {
    if (!cm || (mode != FFH_EXCHANGE_GATHER && mode != FFH_EXCHANGE_SLICE)) return FFH_E_ARG;
    cm->exchange = mode;   // (every rank of the communicator must choose the same form: the collectives differ)
    return FFH_OK;
}
int ffh_comm_get_exchange(const ffh_comm *cm) { return cm ? cm->exchange : FFH_E_ARG; }
int ffh_comm_timings(const ffh_comm *cm, double *scan_ms, double *exchange_ms) {
    if (!cm) return FFH_E_ARG;
    if (scan_ms) *scan_ms = cm->scan_ms;
    if (exchange_ms) *exchange_ms = cm->exchange_ms;
    return FFH_OK;
}

// The exchange of the shards' aggregates after every local shard has been scanned: stream-ordered on the contexts' streams.
//   every shard aggregates as if it were the first one (prior 0: exact for every guide whose cut-off the shards before it do not move)
//   -> ONE all-gather of the 88-byte summaries + a status record per shard
//   -> every rank folds all shards' records itself, in shard order (k_exchange_reduce), which also yields its own prior
//   -> one polled host wait: did any guide's cut-off fall inside a shard with a non-zero prior?  (None in a guide set without
//      OVERFLOW guides: done.)  If so: those shards aggregate the affected guides again with their prior, a second all-gather, a second fold.
// Round 3 issued four collectives per step (totals all-gather, MAX all-reduce, SUM all-reduce, f64 all-gather) and two epilogue passes
// per shard whatever the guides; on xGMI every collective is a latency.  local_rc[i] != 0: shard i could not be scanned -- it still
// takes part (ranks must not be left waiting inside a collective) and every rank returns the failure.
static int comm_exchange(ffh_comm *cm, uint32_t G, int max_ot, unsigned flags, ffh_guide_summary *summaries_out, const int *local_rc = nullptr) {
    const size_t L = cm->ctx.size();
    const uint32_t W = (uint32_t)cm->world;
    const unsigned jost = flags & FFH_FINALIZE_JOST;
    // the exchange by guide slices (see k_exchange_reduce_slice): sl guides per rank; one rank or no guide: the all-gather form is the same thing
    const bool slice = cm->exchange == FFH_EXCHANGE_SLICE && W > 1 && G > 0;
    const uint32_t sl = slice ? (G + W - 1) / W : 0u;
    std::vector<int> ok(L, 1);
    for (size_t i = 0; i < L; ++i) {
        ffh_ctx *ctx = cm->ctx[i];
        ffh_comm::Buf &b = *cm->buf[i];
        FFC_HIP(hipSetDevice(ctx->device));
        FFC_HIP(b.totals.reserve((size_t)G + 1)); FFC_HIP(b.prior.reserve((size_t)G + 1)); FFC_HIP(b.flag.reserve(4));
        FFC_HIP(b.summ.reserve((size_t)G + 1)); FFC_HIP(b.all.reserve((size_t)W * ((size_t)G + 1))); FFC_HIP(b.red.reserve(std::max((size_t)G, (size_t)W * sl) + 1));
        if (slice) {
            FFC_HIP(b.send.reserve((size_t)W * (sl + 1))); FFC_HIP(b.red_slice.reserve((size_t)sl + 1));
            FFC_HIP(b.prior_all.reserve((size_t)W * (sl + 1))); FFC_HIP(b.prior_in.reserve((size_t)W * (sl + 1)));
        }
        FFC_HIP(ctx->n_ret.reserve((size_t)G + 1));
        if (local_rc && local_rc[i]) ok[i] = 0;
        else if (!ctx->scanned || ctx->n_guides != G) { ok[i] = 0; ctx->err = "the shard has not been scanned with this guide set"; }
        // a scan bounded by a smaller limit than this exchange asks for lacks hits: scan again, unbounded (as ffh_finalize does)
        else if (check_bound(ctx, max_ot) != FFH_OK) ok[i] = 0;
    }
    auto epilogue = [&](size_t i, const uint32_t *d_prior, const uint32_t *d_fix, uint32_t *d_totals) {
        ffh_ctx *ctx = cm->ctx[i];
        (void)hipSetDevice(ctx->device);
        if (G) hipLaunchKernelGGL(k_guide_epilogue, dim3(blocks_for(G, 4)), dim3(256), 0, ctx->st, ctx->seg_begin.p, ctx->seg_end.p,
                                  (const uint64_t *)(ctx->hit_t_ready ? ctx->hit_t.p : nullptr), (const uint64_t *)ctx->hits_sorted, (const uint64_t *)ctx->targets.p, ctx->tbits,
                                  d_prior, ctx->guides.p, ctx->geo, ctx->d_tab, G, (uint32_t)max_ot, jost ? 1 : 0, ctx->n_ret.p, cm->buf[i]->summ.p, d_totals, d_fix,
                                  (GuideSummary *)nullptr);
    };
    constexpr uint64_t kWords = sizeof(GuideSummary) / 8;
    auto gather = [&]() -> int {
        if (slice) {   // every shard packs its records by destination, then the all-to-all: shard j receives everybody's records of slice j
            for (size_t i = 0; i < L; ++i) {
                FFC_HIP(hipSetDevice(cm->ctx[i]->device));
                hipLaunchKernelGGL(k_slice_pack, dim3(blocks_for((uint64_t)W * (sl + 1), 256)), dim3(256), 0, cm->ctx[i]->st, (const GuideSummary *)cm->buf[i]->summ.p, G, sl, W, cm->buf[i]->send.p);
            }
            FFC_HIP(hipGetLastError());
            return comm_all_to_all<uint64_t>(cm, ((uint64_t)sl + 1) * kWords, ncclUint64, [&](size_t i) { return (const uint64_t *)cm->buf[i]->send.p; },
                                             [&](size_t j) { return (uint64_t *)cm->buf[j]->all.p; });
        }
        return comm_all_gather<uint64_t>(cm, ((uint64_t)G + 1) * kWords, ncclUint64, [&](size_t i) { return (const uint64_t *)cm->buf[i]->summ.p; },
                                         [&](size_t j) { return (uint64_t *)cm->buf[j]->all.p; });
    };
    auto reduce = [&](int adjusted) -> int {
        if (slice) {
            for (size_t i = 0; i < L; ++i) {
                ffh_comm::Buf &b = *cm->buf[i];
                const uint64_t s0 = (uint64_t)(cm->first + (int)i) * sl;
                const uint32_t n_slice = s0 >= G ? 0u : (uint32_t)std::min<uint64_t>(sl, G - s0);
                FFC_HIP(hipSetDevice(cm->ctx[i]->device));
                FFC_HIP(hipMemsetAsync(b.flag.p, 0, 8, cm->ctx[i]->st));
                hipLaunchKernelGGL(k_exchange_reduce_slice, dim3(blocks_for((uint64_t)sl + 1, 256)), dim3(256), 0, cm->ctx[i]->st, (const GuideSummary *)b.all.p, n_slice, sl, W, (uint32_t)max_ot,
                                   adjusted, b.prior_all.p, b.red_slice.p, b.flag.p);
                if (!adjusted) hipLaunchKernelGGL(k_slice_flag, dim3(1), dim3(1024), 0, cm->ctx[i]->st, (const uint32_t *)b.flag.p, sl, W, b.prior_all.p);
            }
            FFC_HIP(hipGetLastError());
            if (adjusted) return FFH_OK;
            // the priors (and the slices' flag words) back to the shards they are about
            const int rc = comm_all_to_all<uint32_t>(cm, (uint64_t)sl + 1, ncclUint32, [&](size_t i) { return (const uint32_t *)cm->buf[i]->prior_all.p; },
                                                     [&](size_t j) { return cm->buf[j]->prior_in.p; });
            if (rc) return rc;
            for (size_t i = 0; i < L; ++i) {
                ffh_comm::Buf &b = *cm->buf[i];
                FFC_HIP(hipSetDevice(cm->ctx[i]->device));
                hipLaunchKernelGGL(k_slice_assemble, dim3(blocks_for((uint64_t)G + 1, 256)), dim3(256), 0, cm->ctx[i]->st, (const uint32_t *)b.prior_in.p, G, sl, W, 1, b.prior.p, b.flag.p);
            }
            FFC_HIP(hipGetLastError());
            return FFH_OK;
        }
        for (size_t i = 0; i < L; ++i) {
            ffh_comm::Buf &b = *cm->buf[i];
            FFC_HIP(hipSetDevice(cm->ctx[i]->device));
            FFC_HIP(hipMemsetAsync(b.flag.p, 0, 8, cm->ctx[i]->st));
            hipLaunchKernelGGL(k_exchange_reduce, dim3(blocks_for((uint64_t)G + 1, 256)), dim3(256), 0, cm->ctx[i]->st, (const GuideSummary *)b.all.p, G, W, (uint32_t)max_ot, adjusted,
                               (uint32_t)(cm->first + (int)i), b.prior.p, b.red.p, b.flag.p);
        }
        FFC_HIP(hipGetLastError());
        return FFH_OK;
    };
    // every shard aggregates as if it were the first one; the same pass yields its saturated totals.  Record G is the status record.
    static_assert(sizeof(GuideSummary) % 8 == 0, "gathered as 64-bit words");
    for (size_t i = 0; i < L; ++i) {
        FFC_HIP(hipSetDevice(cm->ctx[i]->device));
        if (ok[i]) {
            epilogue(i, nullptr, nullptr, cm->buf[i]->totals.p);
            FFC_HIP(hipMemsetAsync(cm->buf[i]->summ.p + G, 0, sizeof(GuideSummary), cm->ctx[i]->st));
        } else {
            FFC_HIP(hipMemsetAsync(cm->buf[i]->summ.p, 0, (size_t)G * sizeof(GuideSummary), cm->ctx[i]->st));
            FFC_HIP(hipMemsetAsync(cm->buf[i]->summ.p + G, cm->ctx[i]->too_many_hits ? 0xFE : 0xFF, sizeof(GuideSummary), cm->ctx[i]->st));
        }
    }
    int rc = gather();
    if (rc) return rc;
    rc = reduce(0);
    if (rc) return rc;
    uint32_t word = 0;
    FFC_HIP(hipSetDevice(cm->ctx[0]->device));
    FFC_HIP(spin_wait(cm->ctx[0], nullptr, cm->buf[0]->flag.p, &word));
    if (word & 0x80000000u) {   // every rank sees the same records, so every rank leaves here
        for (size_t i = 0; i < L; ++i) { (void)hipSetDevice(cm->ctx[i]->device); (void)hipStreamSynchronize(cm->ctx[i]->st); }
        cm->err = "a shard could not be scanned";
        for (size_t i = 0; i < L; ++i) if (!ok[i]) cm->err = "shard " + std::to_string(cm->first + (int)i) + ": " + cm->ctx[i]->err;
        return (word & 0x40000000u) ? FFH_E_STATE : kSplitGuides;
    }
    cm->crossing = word;
    if (word) {
        // ... and only the guides whose cut-off the shards before it move are aggregated again (the kernel leaves the others at once)
        for (size_t i = 0; i < L; ++i) epilogue(i, cm->buf[i]->prior.p, cm->buf[i]->totals.p, nullptr);
        rc = gather();
        if (rc) return rc;
        rc = reduce(1);
        if (rc) return rc;
    }
    if (slice) {   // the folded slices to every rank: red = [world][sl], slice j holds guides j * sl .. -- the reduced array of the all-gather form
        rc = comm_all_gather<uint64_t>(cm, (uint64_t)sl * kWords, ncclUint64, [&](size_t i) { return (const uint64_t *)cm->buf[i]->red_slice.p; },
                                       [&](size_t j) { return (uint64_t *)cm->buf[j]->red.p; });
        if (rc) return rc;
    }
    if (summaries_out && G) {
        FFC_HIP(hipSetDevice(cm->ctx[0]->device));
        FFC_HIP(hipMemcpyAsync(summaries_out, cm->buf[0]->red.p, (size_t)G * sizeof(ffh_guide_summary), hipMemcpyDeviceToHost, cm->ctx[0]->st));
    }
    for (size_t i = 0; i < L; ++i) { FFC_HIP(hipSetDevice(cm->ctx[i]->device)); FFC_HIP(hipStreamSynchronize(cm->ctx[i]->st)); }
    return FFH_OK;
}

static int discover_sharded_once(ffh_comm *cm, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets, unsigned flags, ffh_guide_summary *summaries_out);
// A guide set that collects more raw hits than one scan holds on ANY shard: every rank learns it from the status records of the same
// exchange, so every rank halves the guide set at the same place and runs the halves one after the other (no extra collective).  The
// per-guide summaries of the halves are concatenated; hit lists and device summaries then belong to the last part only, so
// ffh_comm_shard_lists / ffh_comm_device_summaries refuse until a call that was not split.
static int discover_sharded_split(ffh_comm *cm, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets, unsigned flags, ffh_guide_summary *summaries_out,
                                  bool &split) {
    int rc = discover_sharded_once(cm, guides, n_guides, max_mismatch, max_offtargets, flags, summaries_out);
    if (rc != kSplitGuides) return rc;
    if (n_guides < 2) return FFH_E_STATE;
    split = true;
    const uint32_t h = n_guides / 2;
    // (the timings of the attempt that had to be split and of the parts, added up once: every call below overwrites cm->scan_ms / exchange_ms)
    double scan = cm->scan_ms, ex = cm->exchange_ms;
    rc = discover_sharded_split(cm, guides, h, max_mismatch, max_offtargets, flags, summaries_out, split);
    scan += cm->scan_ms; ex += cm->exchange_ms;
    if (!rc) {
        rc = discover_sharded_split(cm, guides + h, n_guides - h, max_mismatch, max_offtargets, flags, summaries_out ? summaries_out + h : nullptr, split);
        scan += cm->scan_ms; ex += cm->exchange_ms;
    }
    cm->scan_ms = scan; cm->exchange_ms = ex;
    if (!rc) cm->err.clear();   // (the "more than 2^32 raw hits" text of the attempt that was split is not this call's outcome)
    return rc;
}
int ffh_discover_sharded(ffh_comm *cm, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets, unsigned flags, ffh_guide_summary *summaries_out) try {
    if (!cm || (n_guides && !guides) || max_mismatch < 0 || max_offtargets < 0) { if (cm) cm->err = "bad argument"; return FFH_E_ARG; }
    bool split = false;
    const int rc = discover_sharded_split(cm, guides, n_guides, max_mismatch, max_offtargets, flags, summaries_out, split);
    cm->was_split = !rc && split;
    if (cm->was_split) cm->exchanged = false;
    return rc;
} FFH_CATCH(cm ? &cm->err : nullptr)
static int discover_sharded_once(ffh_comm *cm, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets, unsigned flags, ffh_guide_summary *summaries_out) {
    const size_t L = cm->ctx.size();
    cm->exchanged = false;
    const auto t0 = std::chrono::steady_clock::now();
    // the scans: every local shard on its own host thread (a scan reads its hit count back before it orders the hits, so one thread
    // would run the GPUs one after the other); each shard is bounded by its own totals (the prior of the lower shards could only
    // retire more guides)
    std::vector<int> rcs(L, FFH_OK);
    auto scan = [&](size_t i) { rcs[i] = scan_retry_bounded(cm->ctx[i], guides, n_guides, max_mismatch, max_offtargets); };
    if (L == 1) scan(0);
    else {
        std::vector<std::thread> th;
        for (size_t i = 1; i < L; ++i) th.emplace_back(scan, i);
        scan(0);
        for (auto &t : th) t.join();
    }
    // (a shard that could not be scanned still takes part in the exchange: the other ranks must not be left inside a collective)
    const auto t1 = std::chrono::steady_clock::now();
    const int rc = comm_exchange(cm, n_guides, max_offtargets, flags, summaries_out, rcs.data());
    cm->scan_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    cm->exchange_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    if (rc) return rc;
    cm->n_guides = n_guides; cm->max_ot = max_offtargets; cm->exchanged = true;
    return FFH_OK;
}

// the exchange alone, for callers that scanned the shards themselves (ffh_scan / ffh_scan_bounded on every local context)
int ffh_comm_exchange(ffh_comm *cm, uint32_t n_guides, int max_offtargets, unsigned flags, ffh_guide_summary *summaries_out) try {
    if (!cm || max_offtargets < 0) { if (cm) cm->err = "bad argument"; return FFH_E_ARG; }
    cm->exchanged = false; cm->was_split = false;
    const auto t1 = std::chrono::steady_clock::now();
    const int rc = comm_exchange(cm, n_guides, max_offtargets, flags, summaries_out);
    if (rc) return rc == kSplitGuides ? FFH_E_STATE : rc;   // (the caller scanned the shards itself: it splits the guide set itself)
    cm->n_guides = n_guides; cm->max_ot = max_offtargets; cm->exchanged = true;
    cm->exchange_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    return FFH_OK;
} FFH_CATCH(cm ? &cm->err : nullptr)

int ffh_comm_shard_lists(ffh_comm *cm, int local_shard, unsigned flags, ffh_result **out) try {
    if (!cm || !out || local_shard < 0 || (size_t)local_shard >= cm->ctx.size()) { if (cm) cm->err = "bad argument"; return FFH_E_ARG; }
    if (!cm->exchanged) { cm->err = cm->was_split ? kSplitMessage : "ffh_discover_sharded has not run"; return FFH_E_STATE; }
    ffh_ctx *ctx = cm->ctx[(size_t)local_shard];
    const int rc = ffh_finalize(ctx, cm->buf[(size_t)local_shard]->prior.p, cm->max_ot, (flags & ~FFH_FINALIZE_SUMMARIES_ONLY) | FFH_FINALIZE_PRIOR_ON_DEVICE, out);
    if (rc) { std::lock_guard<std::mutex> g(cm->err_m); cm->err = ctx->err; }
    return rc;
} FFH_CATCH(cm ? &cm->err : nullptr)

int ffh_comm_device_summaries(ffh_comm *cm, int local_shard, const void **device_summaries) {
    if (!cm || !device_summaries || local_shard < 0 || (size_t)local_shard >= cm->ctx.size()) return FFH_E_ARG;
    if (!cm->exchanged) { cm->err = cm->was_split ? kSplitMessage : "ffh_discover_sharded has not run"; return FFH_E_STATE; }
    *device_summaries = cm->buf[(size_t)local_shard]->red.p;
    return FFH_OK;
}

}  // extern "C"
