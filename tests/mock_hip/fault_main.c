/* tests/mock_hip/fault_main.c -- fault injection over the mock runtime (round 6): ONE scenario through the library's host logic -- context, empty
 * database, discover with lists and aggregates, a shared database, a pipe, a two-shard communicator in both exchange forms, a database file
 * written and opened -- with the n-th HIP call failing (MOCK_HIP_FAIL_AT=n; tests/test_library_cpu.py walks n over the whole scenario).  Whatever
 * fails, the library must return an error code, and after everything that exists has been destroyed nothing may be left allocated and nothing
 * may have been freed twice.  MOCK_HIP_FAIL_AT=0: no fault; prints the number of HIP calls of the scenario. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/flashfry_hip.h"

void mock_hip_counts(long long *out);
long long mock_hip_calls(void);
void mock_hip_dump(void);
__attribute__((weak)) void mock_new_arm(long long n);        /* tests/mock_hip/mock_new.cpp, when it is preloaded too */
__attribute__((weak)) long long mock_new_count(void);

int main(void) {
    uint64_t guides[48];
    for (int i = 0; i < 48; ++i) guides[i] = (1ull << 48) | ((uint64_t)(i * 2654435761u) << 6) | 0x2A;
    int errors = 0, steps = 0;
    if (mock_new_arm) mock_new_arm(getenv("MOCK_NEW_FAIL_AT") ? atoll(getenv("MOCK_NEW_FAIL_AT")) : 0);   /* the n-th operator new from here on throws */
#define STEP(call) do { ++steps; const int rc_ = (call); if (rc_ != FFH_OK) { ++errors; if (getenv("MOCK_HIP_TRACE")) fprintf(stderr, "step %d failed (%d): %s\n", steps, rc_, #call); } } while (0)
    ffh_ctx *ctx = ffh_create(0, 3), *other = NULL, *sh[2] = {NULL, NULL};
    if (ctx) {
        STEP(ffh_db_load_soa(ctx, NULL, 0, NULL, 0, 0));
        ffh_result *r = NULL;
        STEP(ffh_discover(ctx, guides, 48, 4, 40, 0, &r)); ffh_result_free(r); r = NULL;
        STEP(ffh_discover(ctx, guides, 48, 5, 2000, FFH_FINALIZE_SUMMARIES_ONLY | FFH_FINALIZE_JOST, &r)); ffh_result_free(r); r = NULL;
        STEP(ffh_scan(ctx, guides, 20, 3));
        STEP(ffh_finalize(ctx, NULL, 100, FFH_FINALIZE_NO_POSITIONS, &r)); ffh_result_free(r); r = NULL;
        if (ffh_ctx_share_db(ctx, &other) == FFH_OK) {
            STEP(ffh_discover(other, guides, 30, 4, 60, 0, &r)); ffh_result_free(r); r = NULL;
            ffh_destroy(other);
        } else ++errors;
        ffh_pipe *pipe = NULL;
        if (ffh_pipe_create(ctx, 2, &pipe) == FFH_OK) {
            uint64_t t[6];
            for (int k = 0; k < 6; ++k) if (ffh_pipe_submit(pipe, guides, 8 + k, 4, 2000, k % 2 ? FFH_FINALIZE_SUMMARIES_ONLY : 0, &t[k]) != FFH_OK) t[k] = 0;
            for (int k = 0; k < 4; ++k) if (t[k]) { if (ffh_pipe_wait(pipe, t[k], &r) != FFH_OK) ++errors; ffh_result_free(r); r = NULL; }
            ffh_pipe_destroy(pipe);   /* two tickets left uncollected */
        } else ++errors;
    } else ++errors;
    for (int i = 0; i < 2; ++i) { sh[i] = ffh_create(0, 3); if (sh[i]) STEP(ffh_db_load_soa(sh[i], NULL, 0, NULL, 0, 0)); else ++errors; }
    if (sh[0] && sh[1]) {
        ffh_comm *comm = NULL;
        if (ffh_comm_create_local(sh, 2, &comm) == FFH_OK) {
            struct ffh_guide_summary *out = (struct ffh_guide_summary *)ffh_host_alloc(48 * sizeof *out);
            for (int mode = 0; mode < 2; ++mode) {
                ffh_comm_set_exchange(comm, mode);
                STEP(ffh_discover_sharded(comm, guides, 48, 4, 40, 0, out));
                ffh_result *lists = NULL;
                if (ffh_comm_shard_lists(comm, 1, 0, &lists) != FFH_OK) ++errors;
                ffh_result_free(lists);
            }
            ffh_host_free(out);
            ffh_comm_destroy(comm);
        } else ++errors;
    }
    for (int i = 0; i < 2; ++i) ffh_destroy(sh[i]);
    {
        enum { T = 3000 };
        static uint64_t t[T], p[T];
        uint64_t x = 777;
        for (int i = 0; i < T; ++i) { x += 1 + (x * 2654435761u) % 300000000ull; t[i] = (((x & 0xFFFFFFFFFFull) << 6) | 0x2A) | (1ull << 48); p[i] = (23ull << 52) | (1ull << 32) | (uint64_t)i; }
        const char *contigs[] = {"c1"};
        const char *path = getenv("FFH_MOCK_DB") ? getenv("FFH_MOCK_DB") : "/tmp/ffh_mock_fault_db";
        if (ffh_db_write(path, 3, 7, contigs, 1, t, T, p, T) == FFH_OK && ctx) {
            STEP(ffh_db_open(ctx, path, 0, 0));
            const char *modes[][2] = {{"FFH_INFLATE", "host"}, {"FFH_LOAD_PIPELINE", "1"}};   /* the loaders with host threads of their own (read when a context is created) */
            for (int m = 0; m < 2; ++m) {
                setenv(modes[m][0], modes[m][1], 1);
                ffh_ctx *c2 = ffh_create(0, 0);
                if (c2) { STEP(ffh_db_open(c2, path, 0, 0)); ffh_destroy(c2); } else ++errors;
                unsetenv(modes[m][0]);
            }
        }
    }
    /* ---- the rest of the ABI's entry points that allocate or copy: score, bulge search, shard totals, the two-step shard epilogue, the indexer ---- */
    if (ctx) {
        ffh_result *r = NULL;
        uint64_t offs[4] = {0, 2, 2, 5}, tg[5];
        for (int i = 0; i < 5; ++i) tg[i] = guides[i] ^ (3ull << (6 + 2 * i));
        STEP(ffh_score_lists(ctx, guides, 3, offs, tg, &r)); ffh_result_free(r); r = NULL;
        STEP(ffh_db_load_soa(ctx, NULL, 0, NULL, 0, 0));
        ffh_ctx *cpf1 = ffh_create(0, 1);   /* (the bulge search is Cas12a's) */
        if (cpf1) {
            ffh_bulge_result *br = NULL;
            STEP(ffh_db_load_soa(cpf1, NULL, 0, NULL, 0, 0));
            STEP(ffh_discover_bulge(cpf1, guides, 16, 3, 1, 0, &br)); ffh_bulge_result_free(br); br = NULL;
            STEP(ffh_discover_bulge(cpf1, guides, 16, 2, 1, FFH_BULGE_PAM_TTTV | FFH_BULGE_BRUTE_FORCE, &br)); ffh_bulge_result_free(br);
            ffh_destroy(cpf1);
        } else ++errors;
        STEP(ffh_scan_bounded(ctx, guides, 24, 4, 40));
        uint32_t totals[24];
        STEP(ffh_shard_totals(ctx, totals, 40));
        void *dsum = ffh_host_alloc(24 * sizeof(struct ffh_guide_summary)), *dtot = ffh_host_alloc(24 * 4);   /* (the mock's device memory is host memory) */
        if (dsum && dtot) {
            STEP(ffh_finalize_shard(ctx, 40, 0, dsum, (uint32_t *)dtot));
            STEP(ffh_exchange_prior(ctx, (const uint32_t *)dtot, 24, 0, 40, (uint32_t *)dtot));
            STEP(ffh_finalize_shard_fixup(ctx, 40, 0, (const uint32_t *)dtot, (const uint32_t *)dtot, dsum));
            STEP(ffh_summaries_to_device(ctx, dsum));
        }
        ffh_host_free(dsum); ffh_host_free(dtot);
        int64_t blocks[3] = {1, 1, 1};     /* three empty linear blocks (BlockManager.scala:431) */
        uint64_t boffs[4] = {0, 1, 2, 3};
        STEP(ffh_db_load_blocks(ctx, blocks, boffs, 3));
    }
    {
        ffh_indexer *ix = ffh_indexer_create(0, 3);
        if (ix) {
            static char seq[20000];
            for (int i = 0; i < 20000; ++i) seq[i] = "ACGT"[(i * 7 + i / 3) & 3];
            STEP(ffh_indexer_add_contig(ix, "chrA", seq, sizeof seq));
            STEP(ffh_indexer_add_contig(ix, "chrB", seq, 5000));
            const char *path = getenv("FFH_MOCK_DB") ? getenv("FFH_MOCK_DB") : "/tmp/ffh_mock_fault_db";
            char ipath[600];
            snprintf(ipath, sizeof ipath, "%s.indexed", path);
            ffh_index_stats ist;
            STEP(ffh_indexer_finish(ix, ipath, 7, &ist));
            ffh_indexer_destroy(ix);
        } else ++errors;
    }
    ffh_destroy(ctx);
    long long c[10];
    mock_hip_counts(c);
    printf("calls %lld steps %d errors %d live %lld bad_frees %lld events %lld/%lld news %lld\n", mock_hip_calls(), steps, errors, c[5], c[4], c[8], c[9], mock_new_count ? mock_new_count() : 0);
    if (c[5]) mock_hip_dump();
    return (c[5] == 0 && c[4] == 0 && c[8] == c[9]) ? 0 : 1;
}
