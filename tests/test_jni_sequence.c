/*
 * test_jni_sequence.c -- the exact call sequence jni/flashfry_jni.c + jni/GPUTraverser.scala make for one Traverser.scan, without a
 * JVM: create(device, 0 = enzyme from the header) -> dbOpen(path, 0, 0) -> discover(guides, maxMismatch, maxOffTargets) ->
 * resultOffsets / resultTargets / resultPosOffsets / resultPositions -> replay per guide in database order -> resultFree -> destroy.
 * The replayed updateOT stream is compared with the CPU oracle's discover on the SAME database file (oracle/ff_oracle_io.c reads the
 * reference format): same hits per guide, same order, same positions, same currentTotal / full.
 *
 *   usage: test_jni_sequence <database path> <guides file: one decimal uint64 per line> <maxMismatch> <maxOffTargets>
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

int main(int argc, char **argv) {
    if (argc != 5) FAIL("usage: %s <db> <guides.txt> <maxMismatch> <maxOffTargets>", argv[0]);
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

    /* ---- GPUTraverser.scan ---- */
    ffh_ctx *ctx = ffh_create(0, 0);                                   /* create(device, enzymeIndex): 0 = take it from the header */
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
