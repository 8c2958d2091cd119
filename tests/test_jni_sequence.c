/*
 * test_jni_sequence.c -- the exact call sequence jni/flashfry_jni.c + jni/GPUTraverser.scala make for one Traverser.scan, without a
 * JVM: create(device, 0 = enzyme from the header) -> dbOpen(path, 0, 0) -> discover(guides, maxMismatch, maxOffTargets) ->
 * resultOffsets / resultTargets / resultPosOffsets / resultPositions -> replay per guide in database order -> resultFree -> destroy.
 * The replayed updateOT stream is compared with the CPU oracle's discover on the SAME database file (oracle/ff_oracle_io.c reads the
 * reference format): same hits per guide, same order, same positions, same currentTotal / full.
 *
 * With a device list (GPUTraverser.devices, -Dflashfry.gpu.devices) the sharded sequence is replayed instead:
 *     create x N -> dbOpenHeader + dbBins + dbBinBytes -> binCuts -> dbOpen(ctx_i, path, cut_i, cut_i+1) -> createLocalComm -> discoverSharded
 *     -> for every shard in order: shardLists -> result* -> replay -> resultFree; commDestroy; destroy x N
 * and the per-guide stream -- shard 0's hits of the guide, then shard 1's, ... -- must again be the oracle's list.  A device named
 * several times (0,0,0) runs the library's copy transport; distinct devices run RCCL.
 *
 *   usage: test_jni_sequence <database path> <guides file: one decimal uint64 per line> <maxMismatch> <maxOffTargets> [devices, e.g. 0,0,0]
 * Built and run by tests/test_jni_binding.py (-m gpu).  TEST INFRASTRUCTURE: links the oracle as the checker.
 */
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/flashfry_hip.h"
#include "../oracle/ff_oracle.h"

#define FAIL(...) do { fprintf(stderr, "FAIL: " __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)

/* GPUTraverser.binCuts */
static void bin_cuts(const uint64_t *bytes, int nbins, int n, int *cut) {
    double total = 0, run = 0;
    for (int b = 0; b < nbins; ++b) total += (double)bytes[b];
    for (int r = 0; r <= n; ++r) cut[r] = nbins;
    cut[0] = 0;
    int r = 1;
    for (int b = 0; b < nbins && r < n; ++b) {
        run += (double)bytes[b];
        while (r < n && run >= total * (double)r / (double)n) cut[r++] = b + 1;
    }
}

/* GPUTraverser.scan with several devices */
static int sharded(const char *db_path, const uint64_t *guides, size_t n, int max_mm, int max_ot, const int *devs, int nd) {
    ffh_ctx *ctx[64];
    for (int i = 0; i < nd; ++i) {
        ctx[i] = ffh_create(devs[i], 0);                                   /* create(devs(i), enzyme) */
        if (!ctx[i]) FAIL("ffh_create: %s", ffh_last_error(NULL));
    }
    if (ffh_db_open_header(ctx[0], db_path)) FAIL("ffh_db_open_header: %s", ffh_last_error(ctx[0]));   /* dbOpenHeader */
    ffh_db_info info;
    if (ffh_db_info_get(ctx[0], &info)) FAIL("ffh_db_info_get");          /* dbBins */
    const int nbins = (int)info.n_bins;
    uint64_t *bytes = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(nbins > 0 ? nbins : 1));
    for (int b = 0; b < nbins; ++b) bytes[b] = ffh_db_bin_bytes(ctx[0], (uint32_t)b);                   /* dbBinBytes */
    int cut[65];
    bin_cuts(bytes, nbins, nd, cut);
    for (int i = 0; i < nd; ++i)
        if (ffh_db_open(ctx[i], db_path, (uint32_t)cut[i], (uint32_t)cut[i + 1])) FAIL("ffh_db_open shard %d: %s", i, ffh_last_error(ctx[i]));   /* dbOpen */
    ffh_comm *comm = NULL;
    if (ffh_comm_create_local(ctx, nd, &comm)) FAIL("ffh_comm_create_local: %s", ffh_comm_last_error(NULL));     /* createLocalComm */
    if (ffh_discover_sharded(comm, guides, (uint32_t)n, max_mm, max_ot, 0u, NULL)) FAIL("ffh_discover_sharded: %s", ffh_comm_last_error(comm));   /* discoverSharded */

    ffo_db *odb = ffo_db_read(db_path);
    if (!odb) FAIL("oracle cannot read the database: %s", ffo_last_error());
    ffo_result *ora = ffo_discover(odb, guides, (int)n, max_mm, max_ot, 0);
    if (!ora) FAIL("oracle discover: %s", ffo_last_error());

    /* the replay: shard by shard, inside a shard guide by guide -- per guide that is database order */
    int *seen = (int *)calloc(n ? n : 1, sizeof(int)), *total = (int *)calloc(n ? n : 1, sizeof(int));
    uint64_t total_hits = 0, overflowed = 0;
    for (int i = 0; i < nd; ++i) {
        ffh_result *res = NULL;
        if (ffh_comm_shard_lists(comm, i, FFH_FINALIZE_NO_HIT_SCORES, &res)) FAIL("ffh_comm_shard_lists %d: %s", i, ffh_comm_last_error(comm));   /* shardLists */
        const uint64_t *off = ffh_result_guide_offsets(res), *tg = ffh_result_hit_targets(res), *po = ffh_result_pos_offsets(res), *ps = ffh_result_positions(res);
        if (ffh_result_n_guides(res) != n) FAIL("n_guides of shard %d", i);
        for (size_t g = 0; g < n; ++g)
            for (uint64_t h = off[g]; h < off[g + 1]; ++h) {
                if (!(total[g] < max_ot || max_ot == 0)) FAIL("guide %zu: a hit of shard %d arrives after the guide is full (addOT's assert)", g, i);
                const int k = seen[g]++;
                if (k >= ffo_result_n_hits(ora, (int)g)) FAIL("guide %zu: more hits than the oracle keeps", g);
                if (tg[h] != ffo_result_hit_target(ora, (int)g, k)) FAIL("guide %zu hit %d (shard %d): target differs", g, k, i);
                const uint64_t np = po[h + 1] - po[h];
                if ((int)np != ffo_result_hit_npos(ora, (int)g, k)) FAIL("guide %zu hit %d: position count differs", g, k);
                if (memcmp(ps + po[h], ffo_result_hit_positions(ora, (int)g, k), np * 8)) FAIL("guide %zu hit %d: positions differ", g, k);
                total[g] += (int)np;
            }
        total_hits += ffh_result_n_hits(res);
        ffh_result_free(res);                                              /* resultFree */
    }
    for (size_t g = 0; g < n; ++g) {
        if (seen[g] != ffo_result_n_hits(ora, (int)g)) FAIL("guide %zu: %d hits over the shards, oracle %d", g, seen[g], ffo_result_n_hits(ora, (int)g));
        if (total[g] != ffo_result_current_total(ora, (int)g)) FAIL("guide %zu: currentTotal", g);
        if ((total[g] >= max_ot) != (ffo_result_full(ora, (int)g) != 0)) FAIL("guide %zu: full", g);
        overflowed += total[g] >= max_ot;
    }
    printf("jni sharded sequence ok: %d shards (transport %d, bin cuts", nd, ffh_comm_transport(comm));
    for (int i = 0; i <= nd; ++i) printf(" %d", cut[i]);
    printf("), %zu guides, %" PRIu64 " hits replayed in shard = database order, %" PRIu64 " guides full, identical to the oracle\n", n, total_hits, overflowed);
    ffo_result_free(ora);
    ffo_db_free(odb);
    ffh_comm_destroy(comm);                                                /* commDestroy */
    for (int i = 0; i < nd; ++i) ffh_destroy(ctx[i]);                      /* destroy */
    free(bytes); free(seen); free(total);
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 5 && argc != 6) FAIL("usage: %s <db> <guides.txt> <maxMismatch> <maxOffTargets> [devices]", argv[0]);
    const char *db_path = argv[1];
    const int max_mm = atoi(argv[3]), max_ot = atoi(argv[4]);
    uint64_t *guides = NULL;
    size_t n = 0, cap = 0;
    FILE *f = fopen(argv[2], "r");
    if (!f) FAIL("cannot open %s", argv[2]);
    uint64_t v;
    while (fscanf(f, "%" SCNu64, &v) == 1) {
        if (n == cap) { cap = cap ? 2 * cap : 1024; guides = (uint64_t *)realloc(guides, cap * sizeof *guides); }
        guides[n++] = v;
    }
    fclose(f);

    int devs[64], nd = 0;
    if (argc == 6) for (char *tok = strtok(argv[5], ","); tok && nd < 64; tok = strtok(NULL, ",")) devs[nd++] = atoi(tok);
    if (nd > 1) return sharded(db_path, guides, n, max_mm, max_ot, devs, nd);

    /* ---- GPUTraverser.scan, one device ---- */
    ffh_ctx *ctx = ffh_create(nd ? devs[0] : 0, 0);                                /* create(device, enzymeIndex): 0 = take it from the header */
    if (!ctx) FAIL("ffh_create: %s", ffh_last_error(NULL));            /* lastError(0) */
    if (ffh_db_open(ctx, db_path, 0, 0)) FAIL("ffh_db_open: %s", ffh_last_error(ctx));   /* dbOpen(ctx, path, 0, 0) */
    ffh_result *res = NULL;
    if (ffh_discover(ctx, guides, (uint32_t)n, max_mm, max_ot, FFH_FINALIZE_NO_HIT_SCORES, &res)) FAIL("ffh_discover: %s", ffh_last_error(ctx));
    const uint64_t *off = ffh_result_guide_offsets(res);               /* resultOffsets */
    const uint64_t *tg = ffh_result_hit_targets(res);                  /* resultTargets */
    const uint64_t *po = ffh_result_pos_offsets(res);                  /* resultPosOffsets */
    const uint64_t *ps = ffh_result_positions(res);                    /* resultPositions */
    if (ffh_result_n_guides(res) != n) FAIL("n_guides");

    /* ---- the checker: the oracle on the same file ---- */
    ffo_db *odb = ffo_db_read(db_path);
    if (!odb) FAIL("oracle cannot read the database: %s", ffo_last_error());
    ffo_result *ora = ffo_discover(odb, guides, (int)n, max_mm, max_ot, 0);
    if (!ora) FAIL("oracle discover: %s", ffo_last_error());

    /* ---- replay, as the Scala loop does: updateOT(guide g, CRISPRHit(tg(h), ps[po(h), po(h + 1)))) ---- */
    uint64_t total_hits = 0, overflowed = 0;
    for (size_t g = 0; g < n; ++g) {
        int current_total = 0;                                          /* CRISPRSiteOT.currentTotal */
        const int want_hits = ffo_result_n_hits(ora, (int)g);
        if ((uint64_t)want_hits != off[g + 1] - off[g]) FAIL("guide %zu: %" PRIu64 " hits, oracle %d", g, off[g + 1] - off[g], want_hits);
        for (uint64_t h = off[g]; h < off[g + 1]; ++h) {
            if (!(current_total < max_ot || max_ot == 0)) FAIL("guide %zu: a hit arrives after the guide is full (addOT's assert)", g);
            const int k = (int)(h - off[g]);
            if (tg[h] != ffo_result_hit_target(ora, (int)g, k)) FAIL("guide %zu hit %d: target differs", g, k);
            const uint64_t np = po[h + 1] - po[h];
            if ((int)np != ffo_result_hit_npos(ora, (int)g, k)) FAIL("guide %zu hit %d: position count differs", g, k);
            if (memcmp(ps + po[h], ffo_result_hit_positions(ora, (int)g, k), np * 8)) FAIL("guide %zu hit %d: positions differ", g, k);
            current_total += (int)np;                                   /* addOT: += offTarget.getOffTargetCount */
        }
        if (current_total != ffo_result_current_total(ora, (int)g)) FAIL("guide %zu: currentTotal", g);
        if ((current_total >= max_ot) != (ffo_result_full(ora, (int)g) != 0)) FAIL("guide %zu: full", g);
        total_hits += off[g + 1] - off[g];
        overflowed += current_total >= max_ot;
    }
    printf("jni sequence ok: %zu guides, %" PRIu64 " hits replayed in database order, %" PRIu64 " guides full, identical to the oracle\n", n, total_hits, overflowed);
    ffo_result_free(ora);
    ffo_db_free(odb);
    ffh_result_free(res);                                               /* resultFree */
    ffh_destroy(ctx);                                                   /* destroy */
    free(guides);
    return 0;
}
