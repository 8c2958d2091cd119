// ffh_api.hip -- context, memory and launch orchestration behind the C ABI of include/flashfry_hip.h.
// gfx950 only.  One context = one GPU + one HIP stream; everything below is issued on that stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
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
#include "ffh_streams.hpp"
#include "ffh_dbfile.hpp"
#include "ffh_ingest.hpp"
#include "ffh_inflate.hpp"
#include "ffh_kernels.hpp"
#include "ffh_compare.hpp"
#include "cfd_table.inc"
#include "jost_table.inc"

// The library is ONE translation unit (every kernel instance is compiled once, every helper is static); its parts, in order:
#include "ffh_ctx.hpp"          // context, device buffers, page-locked result pool, polled host wait
#include "ffh_plan.inc"         // pattern lists, cost model of the prefix / suffix split
#include "ffh_context.inc"      // ffh_version / ffh_last_error / ffh_create / ffh_destroy
#include "ffh_load.inc"         // scan images, ffh_db_load_soa / _blocks / ffh_db_open
#include "ffh_scan.inc"         // candidate lists, bounded scan, hit ordering, ffh_scan*
#include "ffh_finalize.inc"     // ffh_finalize, ffh_discover, ffh_score_lists, result accessors
#include "ffh_share.inc"        // ffh_ctx_share_db: a context that scans another context's resident database through aliases
#include "ffh_pipe.inc"         // ffh_pipe_*: several discover calls in flight against one resident database
#include "ffh_index_api.inc"    // ffh_indexer_*
#include "ffh_bulge_api.inc"    // ffh_discover_bulge
#include "ffh_exchange.inc"     // ffh_finalize_shard, ffh_exchange_*
// =====================================================================================================================
// the bin-sharded discover with the collectives inside the library (RCCL; ffh_comm.hpp)
// =====================================================================================================================
#include "ffh_comm.hpp"
