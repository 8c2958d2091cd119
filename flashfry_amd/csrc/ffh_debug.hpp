// ffh_debug.hpp -- every environment switch of the library's scan path in ONE place, read ONCE when a context is created
// (ffh_create): nothing on a launch path calls getenv.  Most are test hooks and A/B aids (they force a code path that the library
// otherwise picks from the workload); two are resource knobs a deployment may set (FFH_PINNED_LIMIT_MB, FFH_COMPARE_GRID).
// The file-I/O side keeps its own, read where a file is opened or written: FFH_LOAD_THREADS, FFH_VERBOSE (ffh_dbfile.cpp),
// FFH_DEFLATE_LEVEL (ffh_dbwrite.cpp); the communicator reads FFH_RCCL_LIBRARY once per process and FFH_COMM when it is created
// (ffh_comm.hpp); the stream pool reads FFH_STREAM_DESTROY once per process (ffh_streams.hpp).
#pragma once
#include <stdint.h>

#include <cstdlib>
#include <cstring>

namespace ffh {

struct Switches {
    // resource knobs
    long pinned_limit_mb = 0;        // FFH_PINNED_LIMIT_MB: the most page-locked host memory ONE result block may take (0: no limit)
    unsigned compare_grid = 0;       // FFH_COMPARE_GRID: blocks of the compare launch (0: four per CU)
    // test hooks / A-B aids
    bool pool_debug = false;         // FFH_POOL_DEBUG=1: canaries and poison in the page-locked result blocks (ffh_debug_pool_errors)
    bool no_direct = false;          // FFH_NO_DIRECT=1: prefix images keep the slot -> index array (round-2 layout)
    bool no_spin = false;            // FFH_NO_SPIN=1: hipStreamSynchronize instead of polling the published counters
    uint32_t max_guide_batch = 0;    // FFH_MAX_GUIDE_BATCH: guides per compare launch at most (0: what the candidate list allows)
    bool inflate_host = false;       // FFH_INFLATE=host: BGZF members inflated on host threads instead of on the device
    bool inflate_device = false;     // FFH_INFLATE=device: on the device even for a small body (default there: host threads + one copy)
    long work_list_limit = 0;        // FFH_WORK_LIST_LIMIT: first size of the compare launch's work list (forces the run-again path)
    bool slab_prefix_per_slab = false;   // FFH_SLAB_PREFIX=per-slab: a bounded scan bins the prefix candidates per slab
    bool graph = true;               // FFH_GRAPH=0: never replay the candidate-list launches as a captured graph
    int sort_mode = 0;               // FFH_SORT=lsd (1) / seg (2) / bin (3): force one of the hit orderings (0: by the number of hits)
    bool summary_copy = false;       // FFH_SUMMARY_COPY=1: the summaries leave in a copy after the epilogue instead of under it
    bool generic_compare = false;    // FFH_GENERIC_COMPARE=1: the per-width-pair instances of k_compare for every plan
    int work_queue = -1;             // FFH_WORK_QUEUE=0 / 1 / 16 / 4: how the compare launch deals its work entries (-1: by list length)
                                     // (the three tuned instances only: the per-width-pair instances of other plans -- 19-mers, Cpf1,
                                     // <= 2 or >= 6 mismatches -- always deal with a fixed stride, launch_compare_pair)
    bool pipeline = false;           // FFH_PIPELINE=1: a list-delivering ffh_discover scans its guide set in two parts, the first part's lists crossing the
                                     // link under the second part's scan.  OFF: measured slower on this stack (profiles/r05/ab_log.txt 7); kept for tests / A-B
    bool list_zero_copy = false;     // FFH_LIST_ZERO_COPY=1: a list-delivering finalize stores the per-hit arrays and the positions straight into the result's page-locked
                                     // block from the kernels that produce them (as the summaries always are) instead of copying them afterwards.  OFF until measured
                                     // (round 6: built while the GPU pool was closed; results must be the same bytes: tests/test_zz_r6_zero_copy.py)
    bool load_pipeline = false;      // FFH_LOAD_PIPELINE=1: ffh_db_open moves even a small body through the threaded page-locked pipeline (A/B)
    bool slab_totals_sorted = false; // FFH_SLAB_TOTALS=sort: a bounded scan adds up a slab's positions per guide from the records ordered by guide (round 4) instead of k_slab_totals
    bool slab_filter = true;         // FFH_SLAB_FILTER=0: a bounded scan keeps every record of the slab in which a guide reaches the limit (round 4)
    bool one_sweep = false;          // FFH_ONESWEEP=1: the device-wide LSD sort with decoupled look-back instead of a histogram launch + scan per pass.
                                     // OFF: slower on this part (3.27 against 2.45 ms for 4.7e7 keys, profiles/r05/ab_log.txt 8); kept for tests / A-B
    int nb_force[2] = {0, 0};        // FFH_NB_PREFIX / FFH_NB_SUFFIX: buckets per work entry of the image (A/B; 0: side_plan's rule)
    uint64_t raw_hit_limit = (1ull << 32) - 64;   // FFH_RAW_HIT_LIMIT: raw hits one scan may collect before the guide set is split (tests: 2^20)

    static Switches from_env() {
        Switches s;
        auto num = [](const char *name, long dflt) { const char *e = std::getenv(name); return e ? std::atol(e) : dflt; };
        auto is = [](const char *name, const char *v) { const char *e = std::getenv(name); return e && std::strcmp(e, v) == 0; };
        s.pinned_limit_mb = num("FFH_PINNED_LIMIT_MB", 0);
        { const long v = num("FFH_COMPARE_GRID", 0); s.compare_grid = v > 0 ? (unsigned)v : 0u; }
        s.pool_debug = num("FFH_POOL_DEBUG", 0) == 1;
        s.no_direct = num("FFH_NO_DIRECT", 0) == 1;
        s.no_spin = num("FFH_NO_SPIN", 0) == 1;
        { const long v = num("FFH_MAX_GUIDE_BATCH", 0); s.max_guide_batch = v > 0 ? (uint32_t)v : 0u; }
        s.inflate_host = is("FFH_INFLATE", "host");
        s.inflate_device = is("FFH_INFLATE", "device");
        { const long v = num("FFH_WORK_LIST_LIMIT", 0); s.work_list_limit = v > 0 ? v : 0; }
        s.slab_prefix_per_slab = is("FFH_SLAB_PREFIX", "per-slab");
        s.graph = num("FFH_GRAPH", 1) != 0;
        s.sort_mode = is("FFH_SORT", "lsd") ? 1 : is("FFH_SORT", "seg") ? 2 : is("FFH_SORT", "bin") ? 3 : 0;
        s.summary_copy = num("FFH_SUMMARY_COPY", 0) == 1;
        s.generic_compare = num("FFH_GENERIC_COMPARE", 0) == 1;
        s.work_queue = (int)num("FFH_WORK_QUEUE", -1);
        s.pipeline = num("FFH_PIPELINE", 0) == 1;
        s.one_sweep = num("FFH_ONESWEEP", 0) == 1;
        s.slab_filter = num("FFH_SLAB_FILTER", 1) != 0;
        s.slab_totals_sorted = is("FFH_SLAB_TOTALS", "sort");
        s.load_pipeline = num("FFH_LOAD_PIPELINE", 0) == 1;
        s.list_zero_copy = num("FFH_LIST_ZERO_COPY", 0) == 1;
        s.nb_force[0] = (int)num("FFH_NB_PREFIX", 0); s.nb_force[1] = (int)num("FFH_NB_SUFFIX", 0);
        { const long v = num("FFH_RAW_HIT_LIMIT", 0); if (v > 0) s.raw_hit_limit = (uint64_t)v; }
        return s;
    }
};

}  // namespace ffh
