/* tests/mock_hip/host_logic_main.c -- the library's HOST logic end to end on a box without a GPU, over tests/mock_hip/libmock_hip.so (LD_PRELOAD:
 * device memory is host memory, kernels are counted and never run, every count the host reads back is zero).  Round 6's host-side additions:
 *   * the stream pool: fifty contexts created and destroyed -> two streams created, NONE destroyed (csrc/ffh_streams.hpp);
 *   * ffh_ctx_share_db: aliases are never freed twice, the owner refuses to load while it is shared, a sharing context refuses to load or to be
 *     shared, both scan (ffh_discover returns; with kernels that do not run every guide has zero hits), the owner loads again afterwards;
 *   * ffh_pipe_*: three lanes, sixty batches, out-of-order collection, teardown with batches queued;
 *   * ffh_discover_sharded over the copy transport with 2 and 5 shards, both forms of the exchange (all-gather / by guide slices);
 *   * ffh_db_write + ffh_db_open through the three loaders (device inflate, host-thread inflate, the threaded page-locked pipeline);
 *   * at the end: no device or page-locked allocation left, no free of anything that was not allocated.
 * Run by tests/test_library_cpu.py with FFH_NO_SPIN=1 (the polled wait would wait for a kernel that never runs). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/flashfry_hip.h"

void mock_hip_counts(long long *out);
static int bad;
#define EXPECT(c, what) do { if (!(c)) { printf("FAILED: %s\n", what); ++bad; } } while (0)

int main(void) {
    long long c[10];
    EXPECT(ffh_device_count() == 1, "the mock device is visible");
    uint64_t guides[64];
    for (int i = 0; i < 64; ++i) guides[i] = (1ull << 48) | ((uint64_t)(i * 2654435761u) << 6) | 0x2A;
    /* ---- the stream pool ---- */
    for (int k = 0; k < 50; ++k) {
        ffh_ctx *ctx = ffh_create(0, 3);
        EXPECT(ctx != NULL, "ffh_create");
        if (!ctx) return 1;
        EXPECT(ffh_db_load_soa(ctx, NULL, 0, NULL, 0, 0) == FFH_OK, "an empty database loads");
        ffh_result *r = NULL;
        EXPECT(ffh_discover(ctx, guides, 64, 4, 2000, k % 2 ? FFH_FINALIZE_SUMMARIES_ONLY : 0, &r) == FFH_OK && r && ffh_result_n_guides(r) == 64 && ffh_result_n_hits(r) == 0, "discover on it");
        ffh_result_free(r);
        ffh_destroy(ctx);
    }
    mock_hip_counts(c);
    EXPECT(c[0] == 2 && c[1] == 0, "fifty contexts: two streams created, none destroyed");
    EXPECT(c[5] == 0 && c[4] == 0, "nothing left, nothing freed twice after fifty contexts");
    /* ---- a shared database ---- */
    ffh_ctx *owner = ffh_create(0, 3), *other = NULL, *third = NULL;
    EXPECT(ffh_ctx_share_db(owner, &other) == FFH_E_STATE, "nothing to share before a database is loaded");
    EXPECT(ffh_db_load_soa(owner, NULL, 0, NULL, 0, 0) == FFH_OK, "owner loads");
    EXPECT(ffh_ctx_share_db(owner, &other) == FFH_OK && other, "share");
    EXPECT(ffh_db_load_soa(owner, NULL, 0, NULL, 0, 0) == FFH_E_STATE && strstr(ffh_last_error(owner), "shared"), "the owner refuses to load while shared");
    EXPECT(ffh_set_plan(owner, 10, 2) == FFH_OK, "a plan that rebuilds nothing (no targets) is accepted");
    EXPECT(ffh_db_load_soa(other, NULL, 0, NULL, 0, 0) == FFH_E_STATE, "a sharing context cannot load");
    EXPECT(ffh_ctx_share_db(other, &third) == FFH_E_STATE, "a sharing context cannot be shared");
    for (int k = 0; k < 6; ++k) {
        ffh_result *a = NULL, *b = NULL;
        EXPECT(ffh_discover(owner, guides, 64, 3 + k % 3, 40, 0, &a) == FFH_OK && ffh_discover(other, guides, 32, 3 + k % 3, 40, FFH_FINALIZE_NO_POSITIONS, &b) == FFH_OK, "both scan");
        ffh_result_free(a); ffh_result_free(b);
    }
    ffh_destroy(other);
    EXPECT(ffh_db_load_soa(owner, NULL, 0, NULL, 0, 0) == FFH_OK, "not shared any more: the owner loads again");
    /* ---- the pipe ---- */
    ffh_pipe *pipe = NULL;
    EXPECT(ffh_pipe_create(owner, 3, &pipe) == FFH_OK && ffh_pipe_lanes(pipe) == 3, "pipe of three lanes");
    uint64_t ticket[60];
    for (int k = 0; k < 60; ++k) EXPECT(ffh_pipe_submit(pipe, guides, 1 + k % 64, 4, 2000, k % 2 ? FFH_FINALIZE_SUMMARIES_ONLY : 0, &ticket[k]) == FFH_OK, "submit");
    for (int k = 59; k >= 0; --k) {
        ffh_result *r = NULL;
        EXPECT(ffh_pipe_wait(pipe, ticket[k], &r) == FFH_OK && r && ffh_result_n_guides(r) == (uint32_t)(1 + k % 64), "every ticket its own batch");
        ffh_result_free(r);
    }
    for (int k = 0; k < 10; ++k) ffh_pipe_submit(pipe, guides, 8, 4, 2000, 0, &ticket[k]);   /* left uncollected */
    ffh_pipe_destroy(pipe);
    ffh_destroy(owner);
    /* ---- the sharded discover over the copy transport, both forms of the exchange (the host side: buffers, events, rounds, teardown) ---- */
    for (int world = 2; world <= 5; world += 3) {
        ffh_ctx *sh[5];
        for (int i = 0; i < world; ++i) { sh[i] = ffh_create(0, 3); EXPECT(sh[i] && ffh_db_load_soa(sh[i], NULL, 0, NULL, 0, 0) == FFH_OK, "shard context"); }
        ffh_comm *comm = NULL;
        EXPECT(ffh_comm_create_local(sh, world, &comm) == FFH_OK && ffh_comm_transport(comm) == 0 && ffh_comm_world(comm) == world, "communicator over the copy transport");
        struct ffh_guide_summary *out = (struct ffh_guide_summary *)ffh_host_alloc(64 * sizeof *out);
        for (int mode = 0; mode < 2; ++mode) {
            EXPECT(ffh_comm_set_exchange(comm, mode) == FFH_OK && ffh_comm_get_exchange(comm) == mode, "exchange form");
            for (uint32_t n = 1; n <= 64; n += 21) {
                EXPECT(ffh_discover_sharded(comm, guides, n, 4, 40, FFH_FINALIZE_JOST, out) == FFH_OK, "sharded discover");
                ffh_result *lists = NULL;
                EXPECT(ffh_comm_shard_lists(comm, world - 1, FFH_FINALIZE_NO_HIT_SCORES, &lists) == FFH_OK && lists && ffh_result_n_guides(lists) == n, "a shard's lists");
                ffh_result_free(lists);
            }
        }
        EXPECT(ffh_comm_set_exchange(comm, 2) == FFH_E_ARG, "unknown exchange form refused");
        ffh_host_free(out);
        ffh_comm_destroy(comm);
        for (int i = 0; i < world; ++i) ffh_destroy(sh[i]);
    }
    /* ---- the database file path: ffh_db_write (host only: real BGZF + header), then ffh_db_open through every loader -- device inflate (the
     * members' bytes copied as they are), host-thread inflate (zlib + CRC-32 on the loader threads), the threaded page-locked pipeline with its
     * pooled streams.  The decode kernels do not run, so the database that comes up is empty; every host thread, buffer and stream is real. ---- */
    {
        enum { T = 60000 };
        static uint64_t t[T], p[T];
        uint64_t x = 12345;
        for (int i = 0; i < T; ++i) { x += 1 + (x * 2654435761u) % 17000000ull; t[i] = (((x & 0xFFFFFFFFFFull) << 6) | 0x2A) | (1ull << 48); p[i] = (23ull << 52) | (1ull << 32) | (uint64_t)i; }
        const char *contigs[] = {"c1", "c2"};
        const char *path = getenv("FFH_MOCK_DB") ? getenv("FFH_MOCK_DB") : "/tmp/ffh_mock_db";
        EXPECT(ffh_db_write(path, 3, 7, contigs, 2, t, T, p, T) == FFH_OK, "ffh_db_write");
        const char *modes[][2] = {{"FFH_INFLATE", "device"}, {"FFH_INFLATE", "host"}, {"FFH_LOAD_PIPELINE", "1"}};
        for (int m = 0; m < 3; ++m) {
            setenv(modes[m][0], modes[m][1], 1);
            ffh_ctx *ctx = ffh_create(0, 0);
            EXPECT(ctx && ffh_db_open(ctx, path, 0, 0) == FFH_OK, "ffh_db_open");
            EXPECT(ctx && ffh_db_open(ctx, path, 100, 9000) == FFH_OK, "ffh_db_open of a bin range");
            ffh_db_info info;
            EXPECT(ctx && ffh_db_info_get(ctx, &info) == FFH_OK && info.enzyme_index == 3 && info.n_bins == 16384, "the header's enzyme and bins");
            EXPECT(ctx && ffh_db_contig(ctx, 2) && !strcmp(ffh_db_contig(ctx, 2), "c2"), "the contig table");
            ffh_destroy(ctx);
            unsetenv(modes[m][0]);
        }
        EXPECT(ffh_db_open(NULL, path, 0, 0) != FFH_OK, "null context");
        ffh_ctx *ctx = ffh_create(0, 0);
        EXPECT(ffh_db_open(ctx, "/nonexistent/db", 0, 0) == FFH_E_IO, "a missing file is an I/O error");
        ffh_destroy(ctx);
    }
    mock_hip_counts(c);
    EXPECT(c[1] == 0, "no stream destroyed, ever");
    EXPECT(c[0] <= 40, "streams: the pool's handful (two per context alive at once + the loader's lanes)");
    EXPECT(c[5] == 0, "no device or page-locked allocation left behind");
    EXPECT(c[4] == 0, "nothing freed that was not allocated (no alias freed)");
    EXPECT(c[8] == c[9], "every event destroyed");
    printf("%s (streams created %lld destroyed %lld, device allocations %lld, kernel launches counted %lld)\n", bad ? "HOST LOGIC DIFFERENCES" : "host logic over the mock runtime: as specified", c[0], c[1], c[2], c[6]);
    return bad ? 1 : 0;
}
