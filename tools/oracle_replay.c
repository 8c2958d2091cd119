/* tools/oracle_replay.c -- the CHECKER alone on the inputs of sweep cases (tools/oracle_case_dump.py), built under a sanitizer
 * (round 6; test infrastructure like everything that links oracle/).  For every case file: build the database, checksum it, run
 * ffo_discover N times, export every answer and compare it with the first, checksum the database again.  Any difference between two
 * answers, any database change, any sanitizer report is what sweep B's case 1505 (profiles/r05/stress_sweep_b_inproc_4101.log) looked
 * like from outside.
 *
 *   clang -fsanitize=memory -fsanitize-memory-track-origins -g -O1 -Ioracle tools/oracle_replay.c oracle/ff_oracle.c oracle/ff_oracle_score.c oracle/ff_oracle_io.c -lz -lm
 *   gcc   -fsanitize=address,undefined ...            |   plain gcc -O2 + MALLOC_PERTURB_=165 / valgrind
 *   ./a.out [-n repeats] [-d junk_mb] case.bin ...    (-d: malloc/fill/free that much junk between calls, so that later allocations reuse dirty memory) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ff_oracle.h"

static uint64_t fnv(const void *p, size_t n, uint64_t h) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}
static void dirty_heap(size_t mb, unsigned seed) {   /* leave freed chunks of many sizes full of non-zero junk */
    if (!mb) return;
    enum { N = 4096 };
    void *blk[N];
    size_t left = mb << 20;
    int n = 0;
    while (n < N && left > 0) {
        seed = seed * 1664525u + 1013904223u;
        size_t sz = 16 + (seed >> 8) % (n % 7 == 0 ? 1u << 20 : 1u << 12);
        if (sz > left) sz = left;
        blk[n] = malloc(sz);
        memset(blk[n], 0x5A + (n & 31), sz);
        left -= sz;
        n++;
    }
    for (int i = 0; i < n; i += 2) free(blk[i]);
    for (int i = 1; i < n; i += 2) free(blk[i]);
}

int main(int argc, char **argv) {
    int repeats = 3, bad = 0;
    size_t junk_mb = 0;
    int a = 1;
    while (a < argc && argv[a][0] == '-') {
        if (!strcmp(argv[a], "-n")) repeats = atoi(argv[a + 1]);
        else if (!strcmp(argv[a], "-d")) junk_mb = (size_t)atol(argv[a + 1]);
        a += 2;
    }
    for (; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { perror(argv[a]); return 2; }
        uint32_t h4[4]; uint64_t h3[3];
        if (fread(h4, 4, 4, f) != 4 || fread(h3, 8, 3, f) != 3) { fprintf(stderr, "%s: short header\n", argv[a]); return 2; }
        uint64_t *t = (uint64_t *)malloc((h3[0] + 1) * 8), *p = (uint64_t *)malloc((h3[1] + 1) * 8), *g = (uint64_t *)malloc((h3[2] + 1) * 8);
        if (fread(t, 8, h3[0], f) != h3[0] || fread(p, 8, h3[1], f) != h3[1] || fread(g, 8, h3[2], f) != h3[2]) { fprintf(stderr, "%s: short body\n", argv[a]); return 2; }
        fclose(f);
        dirty_heap(junk_mb, 1u);
        ffo_db *db = ffo_db_new((int)h4[0], 7);
        if (!db || ffo_db_build_from_sorted(db, t, p, h3[0], 500)) { fprintf(stderr, "%s: %s\n", argv[a], ffo_last_error()); return 2; }
        const uint64_t sum0 = ffo_db_checksum(db, NULL, NULL, NULL);
        uint64_t first = 0, first_hits = 0;
        for (int r = 0; r < repeats; r++) {
            dirty_heap(junk_mb, 7u + (unsigned)r);
            ffo_result *res = ffo_discover(db, g, (int)h3[2], (int)h4[1], (int)h4[2], 0);
            if (!res) { fprintf(stderr, "%s: discover: %s\n", argv[a], ffo_last_error()); return 2; }
            const size_t H = ffo_result_export(res, NULL, NULL, NULL, NULL), P = ffo_result_total_positions(res);
            uint64_t *goff = (uint64_t *)malloc((h3[2] + 1) * 8), *ht = (uint64_t *)malloc((H + 1) * 8), *po = (uint64_t *)malloc((H + 1) * 8), *ps = (uint64_t *)malloc((P + 1) * 8);
            ffo_result_export(res, goff, ht, po, ps);
            uint64_t h = fnv(goff, (h3[2] + 1) * 8, 1469598103934665603ULL);
            h = fnv(ht, H * 8, h); h = fnv(po, (H + 1) * 8, h); h = fnv(ps, P * 8, h);
            for (int k = 0; k < (int)h3[2]; k++) { int v[3] = {ffo_result_current_total(res, k), ffo_result_full(res, k), ffo_result_n_hits(res, k)}; h = fnv(v, sizeof v, h); }
            if (r == 0) { first = h; first_hits = H; }
            else if (h != first) { printf("%s: ANSWER %d DIFFERS from the first (%zu hits against %llu)\n", argv[a], r, H, (unsigned long long)first_hits); bad++; }
            /* consecutive duplicates inside a guide's list: what case 1505 showed */
            for (int k = 0; k < (int)h3[2]; k++)
                for (uint64_t i = goff[k] + 1; i < goff[k + 1]; i++)
                    if (ht[i] == ht[i - 1]) { printf("%s: answer %d: guide %d lists target %016llx twice\n", argv[a], r, k, (unsigned long long)ht[i]); bad++; break; }
            free(goff); free(ht); free(po); free(ps);
            ffo_result_free(res);
            if (ffo_db_checksum(db, NULL, NULL, NULL) != sum0) { printf("%s: DATABASE CHANGED during answer %d\n", argv[a], r); bad++; }
        }
        printf("%s: enzyme %u mm %u max_ot %u T %llu G %llu: %llu hits, %d answers, %s\n", argv[a], h4[0], h4[1], h4[2], (unsigned long long)h3[0], (unsigned long long)h3[2],
               (unsigned long long)first_hits, repeats, bad ? "DISAGREEMENT" : "all equal");
        ffo_db_free(db);
        free(t); free(p); free(g);
    }
    return bad ? 1 : 0;
}
