/*
 * ff_oracle.c -- CPU oracle (TEST INFRASTRUCTURE; see ff_oracle.h).  Part 1: codecs, blocks, database,
 * traversal, block compare and aggregation.  Plain C99, single-threaded like the reference.
 * Paths in comments are relative to /root/reference/src/main/scala.
 */
#define _GNU_SOURCE
#include "ff_oracle.h"

#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include "ff_oracle_internal.h"

/* ------------------------------------------------------------------------------------------------
 * errors + counters
 * ---------------------------------------------------------------------------------------------- */
static char g_err[512];
const char *ffo_last_error(void) { return g_err; }
void ffo_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

/* per thread: bench.py's all-core baseline runs one discover per thread, and shared counters in the inner loop would make the
 * threads fight over one cache line (the reference itself is single-threaded); initial-exec: a plain %fs-relative access, the
 * default model of a shared object would call __tls_get_addr on every comparison */
static __thread uint64_t g_bit_comparisons __attribute__((tls_model("initial-exec")));  /* BitEncoding.allComparisons, bitcoding/BitEncoding.scala:193 */
static __thread uint64_t g_all_comparisons __attribute__((tls_model("initial-exec")));  /* Traverser.allComparisons, reference/traverser/Traverser.scala:74 */
uint64_t ffo_counter_bit_comparisons(void) { return g_bit_comparisons; }
uint64_t ffo_counter_all_comparisons(void) { return g_all_comparisons; }
void ffo_counters_reset(void) { g_bit_comparisons = g_all_comparisons = 0; }

/* ------------------------------------------------------------------------------------------------
 * parameter packs -- standards/StandardScanParameters.scala:90-215, index map :61-80
 * ---------------------------------------------------------------------------------------------- */
static const ffo_pack PACKS[6] = {
    /* Cpf1ParameterPack :197-214 */
    {1, "CPF1", 24, 4, 1, 0x00FFFFFFFFFFULL, 4, 24, 0},
    /* Cas9ParameterPack :90-109 */
    {2, "SPCAS9", 23, 3, 0, 0x3FFFFFFFFFC0ULL, 0, 20, 1},
    /* Cas9NGGParameterPack :134-153 */
    {3, "SPCAS9NGG", 23, 3, 0, 0x3FFFFFFFFFC0ULL, 0, 20, 1},
    /* Cas9NAGParameterPack :178-197 */
    {4, "SPCAS9NAG", 23, 3, 0, 0x3FFFFFFFFFC0ULL, 0, 20, 1},
    /* Cas9ParameterPack19bp :112-131 */
    {5, "SPCAS919", 22, 3, 0, 0x0FFFFFFFFFC0ULL, 0, 19, 0},
    /* Cas9NGG19ParameterPack :156-175 */
    {6, "SPCAS9NGG19", 22, 3, 0, 0x0FFFFFFFFFC0ULL, 0, 19, 0},
};

const ffo_pack *ffo_pack_by_index(int index) {
    if (index < 1 || index > 6) return NULL;
    return &PACKS[index - 1];
}

const ffo_pack *ffo_pack_by_name(const char *name) { /* nameToParameterPack :51-59 (case-insensitive) */
    for (int i = 0; i < 6; i++)
        if (strcasecmp(name, PACKS[i].name) == 0) return &PACKS[i];
    return NULL;
}

/* ------------------------------------------------------------------------------------------------
 * bitcoding/BitEncoding.scala
 * ---------------------------------------------------------------------------------------------- */
int ffo_bit_encode(const char *s, int len, int count, uint64_t *out) { /* :46-67 */
    if (len > 24) { ffo_set_error("String too long to be encoded (%d > 24)", len); return -1; }  /* :47 */
    if (count < 1) { ffo_set_error("String count <= 0"); return -2; }                            /* :48 */
    uint64_t enc = 0;
    for (int i = 0; i < len; i++) {
        enc <<= 2; /* :53 */
        switch (toupper((unsigned char)s[i])) {
            case 'A': enc |= 0; break;
            case 'C': enc |= 1; break;
            case 'G': enc |= 2; break;
            case 'T': enc |= 3; break;
            default: ffo_set_error("Unable to encode character %c", s[i]); return -3; /* :60 */
        }
    }
    *out = enc | ((uint64_t)(int64_t)count << 48); /* :66 */
    return 0;
}

int ffo_get_count(uint64_t enc) { return (int)(int16_t)(enc >> 48); } /* :114 -- toShort */

uint64_t ffo_update_count(uint64_t enc, int count) { /* :108-111 */
    return (enc & FFO_STRING_MASK) | ((uint64_t)(int64_t)(int16_t)count << 48);
}

int ffo_bit_decode(uint64_t enc, int actual_size, char *out) { /* :85-99 */
    static const char B[4] = {'A', 'C', 'G', 'T'};
    for (int i = 0; i < actual_size; i++) out[actual_size - 1 - i] = B[(enc >> (2 * i)) & 3]; /* built reversed then .reverse */
    out[actual_size] = 0;
    return ffo_get_count(enc);
}

int ffo_mismatches(const ffo_pack *p, uint64_t e1, uint64_t e2, uint64_t additional_mask) { /* :127-132 */
    g_bit_comparisons++;                                                                   /* :128 */
    uint64_t first = (e1 ^ e2) & additional_mask & p->cmp_mask;                             /* :129 */
    return __builtin_popcountll((first & FFO_UPPER_BITS) | ((first << 1) & FFO_UPPER_BITS)); /* :130 */
}

uint64_t ffo_bin_shift(const ffo_pack *p, int bin_size, uint64_t base, int rshift) { /* :179-185 */
    int sh = p->five_prime ? 2 * (p->scan_len - (bin_size + p->pam_len + rshift))
                           : 2 * (p->scan_len - (bin_size + rshift));
    return (base << sh) & FFO_STRING_MASK;
}

uint64_t ffo_comp_bitmask_for_bin(const ffo_pack *p, int bin_size, int rshift) { /* :167-170 */
    uint64_t base = FFO_STRING_MASK >> (48 - bin_size * 2);
    return ffo_bin_shift(p, bin_size, base, rshift);
}

int ffo_bin_to_long_comparitor(const ffo_pack *p, const char *bin, int bin_size, int rshift,
                               ffo_bin_and_mask *out) { /* :153-157 */
    uint64_t enc;
    int rc = ffo_bit_encode(bin, bin_size, 1, &enc); /* bitEncodeString(bin) => count 1 in the top bits ... */
    if (rc) return rc;
    out->bin_long = ffo_bin_shift(p, bin_size, enc, rshift); /* ... which "& stringMask" in binShift removes */
    out->guide_mask = ffo_comp_bitmask_for_bin(p, bin_size, rshift);
    return 0;
}

int ffo_mismatch_bin(const ffo_pack *p, const ffo_bin_and_mask *bin, uint64_t guide) { /* :142-144 */
    return ffo_mismatches(p, bin->bin_long, guide & bin->guide_mask, FFO_STRING_MASK);
}

/* utils/BaseCombinationGenerator.scala:33-69: iteration order AAAA.. -> TTTT.., last position fastest */
void ffo_bin_name(int width, uint32_t idx, char *out) {
    static const char B[4] = {'A', 'C', 'G', 'T'};
    for (int i = 0; i < width; i++) out[i] = B[(idx >> (2 * (width - 1 - i))) & 3];
    out[width] = 0;
}

/* utils/Utils.scala:154-186 -- ByteBuffer in nativeOrder; x86 => little-endian (UtilsTest.scala:38-46) */
void ffo_longs_to_bytes(const int64_t *longs, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        uint64_t v = (uint64_t)longs[i];
        for (int b = 0; b < 8; b++) out[i * 8 + b] = (uint8_t)(v >> (8 * b));
    }
}
void ffo_bytes_to_longs(const uint8_t *bytes, size_t nbytes, int64_t *out) {
    for (size_t i = 0; i < nbytes / 8; i++) {
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v |= (uint64_t)bytes[i * 8 + b] << (8 * b);
        out[i] = (int64_t)v;
    }
}

/* ------------------------------------------------------------------------------------------------
 * bitcoding/BitPosition.scala:51-92
 * ---------------------------------------------------------------------------------------------- */
uint64_t ffo_pos_encode(int contig_id, uint32_t position, int target_len, int forward) { /* :51-63 */
    uint64_t contig = (uint64_t)contig_id << 32;
    uint64_t pos = (uint64_t)position;
    uint64_t strand = forward ? 0ULL : (1ULL << 60); /* :58, precedence quirk yields exactly this */
    uint64_t size = (uint64_t)target_len << 52;
    return contig | pos | strand | size;
}
void ffo_pos_decode(uint64_t enc, int *contig_id, uint32_t *start, int *size, int *forward) { /* :65-72 */
    *contig_id = (int)((enc & 0x000FFFFF00000000ULL) >> 32);
    *start = (uint32_t)(enc & 0x00000000FFFFFFFFULL);
    *size = (int)((enc & 0x0FF0000000000000ULL) >> 52);
    *forward = ((enc & 0xF000000000000000ULL) >> 60) == 0;
}

/* ------------------------------------------------------------------------------------------------
 * block writers -- reference/binary/blocks/BlockManager.scala:362-442
 * ---------------------------------------------------------------------------------------------- */
size_t ffo_create_linear_block(const uint64_t *targets, const uint64_t *positions, size_t n, int64_t *out) { /* :424-442 */
    size_t w = 0, pi = 0;
    if (out) out[w] = 1; /* block ID :431 */
    w++;
    for (size_t i = 0; i < n; i++) {
        int cnt = ffo_get_count(targets[i]);
        if (out) out[w] = (int64_t)targets[i];
        w++;
        for (int k = 0; k < cnt; k++, pi++, w++)
            if (out) out[w] = (int64_t)positions[pi];
    }
    return w;
}

size_t ffo_create_indexed_block(const ffo_pack *p, const uint64_t *targets, const uint64_t *positions, size_t n,
                                int prefix_len, int lookup, int64_t *out) { /* :362-413 */
    int nsub = 1 << (2 * lookup);
    int *first = (int *)malloc(sizeof(int) * nsub), *size = (int *)calloc(nsub, sizeof(int));
    for (int i = 0; i < nsub; i++) first[i] = -1; /* :376-379 */
    size_t cur = 0, pi = 0, w = (size_t)1 + nsub;
    char s[32];
    for (size_t i = 0; i < n; i++) {
        int cnt = ffo_get_count(targets[i]);
        ffo_bit_decode(targets[i], p->scan_len, s);
        /* sub-bin = decoded.slice(prefix.size, prefix.size + lookupBinSize), :384 */
        int sb = 0;
        for (int k = 0; k < lookup; k++) {
            char c = s[prefix_len + k];
            sb = (sb << 2) | (c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3);
        }
        if (first[sb] >= (int)cur || first[sb] < 0) first[sb] = (int)cur; /* :385-387 */
        cur += (size_t)1 + cnt;                                            /* :389 */
        size[sb] += 1 + cnt;                                               /* :390 */
        if (out) out[w] = (int64_t)targets[i];
        w++;
        for (int k = 0; k < cnt; k++, pi++, w++)
            if (out) out[w] = (int64_t)positions[pi];
    }
    if (out) {
        out[0] = 2; /* :397 */
        for (int i = 0; i < nsub; i++) /* (pos.toLong << 32 | size.toLong) :401 -- pos = -1 sign-extends */
            out[1 + i] = (int64_t)(((uint64_t)(int64_t)first[i] << 32) | (uint64_t)(int64_t)size[i]);
    }
    free(first);
    free(size);
    return w;
}

/* ------------------------------------------------------------------------------------------------
 * in-memory database
 * ---------------------------------------------------------------------------------------------- */
ffo_db *ffo_db_new(int enzyme_index, int bin_width) {
    const ffo_pack *p = ffo_pack_by_index(enzyme_index);
    if (!p) { ffo_set_error("Unable to find the correct parameter pack for enzyme: %d", enzyme_index); return NULL; }
    if (bin_width < 1 || bin_width > 12) { ffo_set_error("bad bin width %d", bin_width); return NULL; }
    ffo_db *db = (ffo_db *)calloc(1, sizeof *db);
    db->pack = p;
    db->bin_width = bin_width;
    db->n_bins = 1 << (2 * bin_width);
    db->bins = (ffo_bin *)calloc((size_t)db->n_bins, sizeof(ffo_bin));
    return db;
}
void ffo_db_free(ffo_db *db) {
    if (!db) return;
    if (db->sealed) munmap(db->sealed, db->sealed_bytes);
    else {
        for (int i = 0; i < db->n_bins; i++) free(db->bins[i].longs);
        free(db->bins);
    }
    for (int i = 0; i < db->n_contigs; i++) free(db->contigs[i]);
    free(db->contigs);
    free(db);
}
int ffo_db_seal(ffo_db *db) {
    if (db->sealed) return 0;
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    size_t table = ((sizeof(ffo_bin) * (size_t)db->n_bins + 63) / 64) * 64, longs = 0;
    for (int i = 0; i < db->n_bins; i++) longs += db->bins[i].n_longs;
    size_t bytes = ((table + longs * sizeof(int64_t) + page - 1) / page) * page;
    char *m = (char *)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) { ffo_set_error("ffo_db_seal: mmap of %zu bytes failed", bytes); return -1; }
    ffo_bin *nb = (ffo_bin *)m;
    int64_t *w = (int64_t *)(m + table);
    for (int i = 0; i < db->n_bins; i++) {
        nb[i] = db->bins[i];
        if (db->bins[i].longs) { memcpy(w, db->bins[i].longs, db->bins[i].n_longs * sizeof(int64_t)); nb[i].longs = w; w += db->bins[i].n_longs; free(db->bins[i].longs); }
    }
    free(db->bins);
    db->bins = nb;
    db->sealed = m; db->sealed_bytes = bytes;
    if (mprotect(m, bytes, PROT_READ)) { ffo_set_error("ffo_db_seal: mprotect failed"); return -2; }
    return 0;
}
int ffo_db_set_bin(ffo_db *db, uint32_t bi, const int64_t *longs, size_t n, int n_targets) {
    if ((int)bi >= db->n_bins || db->sealed) return -1;
    free(db->bins[bi].longs);
    db->bins[bi].longs = (int64_t *)malloc(n * sizeof(int64_t));
    memcpy(db->bins[bi].longs, longs, n * sizeof(int64_t));
    db->bins[bi].n_longs = n;
    db->bins[bi].n_targets = n_targets;
    return 0;
}
int ffo_db_add_contig(ffo_db *db, const char *name) { /* BitPosition.addReference, bitcoding/BitPosition.scala:42-49 */
    db->contigs = (char **)realloc(db->contigs, sizeof(char *) * (size_t)(db->n_contigs + 1));
    db->contigs[db->n_contigs++] = strdup(name);
    return db->n_contigs; /* 1-based id */
}
int ffo_db_n_bins(const ffo_db *db) { return db->n_bins; }
/* test aid (tools/stress_parity.py): a checksum over every bin's longs, and the first bin whose own checksum differs from `expect[bin]`
 * (expect == NULL: none).  The randomised sweep asks for it before and after every case: a database whose memory changes under the
 * checker is a heap write by somebody else, a database that stays the same while two discover calls disagree is the checker's own. */
static uint64_t bin_sum(const ffo_bin *b) {
    uint64_t h = 1469598103934665603ULL ^ (uint64_t)b->n_longs;
    for (size_t i = 0; i < b->n_longs; i++) { h ^= (uint64_t)b->longs[i]; h *= 1099511628211ULL; }
    return h;
}
uint64_t ffo_db_checksum(const ffo_db *db, uint64_t *per_bin /* NULL or [n_bins]: filled */, const uint64_t *expect, int *first_changed) {
    uint64_t all = 0;
    if (first_changed) *first_changed = -1;
    for (int i = 0; i < db->n_bins; i++) {
        uint64_t h = db->bins[i].longs ? bin_sum(&db->bins[i]) : 0;
        if (per_bin) per_bin[i] = h;
        if (expect && first_changed && *first_changed < 0 && expect[i] != h) *first_changed = i;
        all = (all ^ h) * 1099511628211ULL;
    }
    return all;
}
int ffo_db_bin_width(const ffo_db *db) { return db->bin_width; }
int ffo_db_enzyme(const ffo_db *db) { return db->pack->index; }
int ffo_db_n_contigs(const ffo_db *db) { return db->n_contigs; }
const char *ffo_db_contig(const ffo_db *db, int id) { return (id >= 1 && id <= db->n_contigs) ? db->contigs[id - 1] : NULL; }
size_t ffo_db_bin_longs(const ffo_db *db, uint32_t bi, const int64_t **longs, int *n_targets) {
    if (longs) *longs = db->bins[bi].longs;
    if (n_targets) *n_targets = db->bins[bi].n_targets;
    return db->bins[bi].n_longs;
}

/* bin index of a target = the bin_width bases that follow the 5' PAM (if any): crispr/BinWriter.scala:58-64,
 * reference/binary/BlockReader.scala:54-81 (selection by mismatches(target, bin, mask) == 0) */
uint32_t ffo_target_bin(const ffo_pack *p, int bin_width, uint64_t target) {
    uint64_t mask = ffo_comp_bitmask_for_bin(p, bin_width, 0);
    int sh = p->five_prime ? 2 * (p->scan_len - (bin_width + p->pam_len)) : 2 * (p->scan_len - bin_width);
    return (uint32_t)((target & mask) >> sh);
}

int ffo_db_build_from_sorted(ffo_db *db, const uint64_t *targets, const uint64_t *positions, size_t n,
                             int max_linear) { /* reference/binary/DatabaseWriter.scala:76-97 */
    const ffo_pack *p = db->pack;
    if (db->sealed) { ffo_set_error("the database is sealed"); return -1; }
    /* group by bin, preserving input order inside each bin (BlockReader.fetchBin keeps sorted order) */
    size_t *bin_n = (size_t *)calloc((size_t)db->n_bins + 1, sizeof(size_t));
    size_t *pos_off = (size_t *)malloc((n + 1) * sizeof(size_t));
    pos_off[0] = 0;
    for (size_t i = 0; i < n; i++) {
        int c = ffo_get_count(targets[i]);
        if (c < 1) { ffo_set_error("target %zu has count %d", i, c); free(bin_n); free(pos_off); return -1; }
        pos_off[i + 1] = pos_off[i] + (size_t)c;
        bin_n[ffo_target_bin(p, db->bin_width, targets[i]) + 1]++;
    }
    for (int b = 0; b < db->n_bins; b++) bin_n[b + 1] += bin_n[b];
    size_t *cursor = (size_t *)malloc(sizeof(size_t) * (size_t)db->n_bins);
    memcpy(cursor, bin_n, sizeof(size_t) * (size_t)db->n_bins);
    size_t *order = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) order[cursor[ffo_target_bin(p, db->bin_width, targets[i])]++] = i;
    for (int b = 0; b < db->n_bins; b++) {
        size_t nb = bin_n[b + 1] - bin_n[b], np = 0;
        for (size_t k = 0; k < nb; k++) np += (size_t)ffo_get_count(targets[order[bin_n[b] + k]]);
        uint64_t *bt = (uint64_t *)malloc(sizeof(uint64_t) * (nb ? nb : 1));
        uint64_t *bp = (uint64_t *)malloc(sizeof(uint64_t) * (np ? np : 1));
        size_t w = 0;
        for (size_t k = 0; k < nb; k++) {
            size_t i = order[bin_n[b] + k];
            bt[k] = targets[i];
            for (size_t q = pos_off[i]; q < pos_off[i + 1]; q++) bp[w++] = positions[q];
        }
        int indexed = ((int)nb > max_linear) && !p->five_prime; /* DatabaseWriter.scala:85 */
        size_t nl = indexed ? ffo_create_indexed_block(p, bt, bp, nb, db->bin_width, 4, NULL)
                            : ffo_create_linear_block(bt, bp, nb, NULL);
        int64_t *blk = (int64_t *)malloc(nl * sizeof(int64_t));
        if (indexed) ffo_create_indexed_block(p, bt, bp, nb, db->bin_width, 4, blk);
        else ffo_create_linear_block(bt, bp, nb, blk);
        free(db->bins[b].longs);
        db->bins[b].longs = blk;
        db->bins[b].n_longs = nl;
        db->bins[b].n_targets = (int)nb;
        free(bt);
        free(bp);
    }
    free(bin_n); free(pos_off); free(cursor); free(order);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * aggregation -- crispr/ResultsAggregator.scala:32-79, crispr/CRISPRSiteOT.scala:31-64
 * ---------------------------------------------------------------------------------------------- */
typedef struct ffo_agg {
    ffo_guide_ot *guides;
    int n;
    /* overflow callback target: traversal state (ResultsAggregator.scala:77) */
    void (*overflow_cb)(void *ctx, int guide_index);
    void *cb_ctx;
} ffo_agg;

static int site_full(const ffo_guide_ot *g) { return g->current_total >= g->overflow; } /* CRISPRSiteOT.scala:39 */

static void site_add_ot(ffo_guide_ot *g, uint64_t target, const uint64_t *positions, int npos) { /* CRISPRSiteOT.scala:41-46 */
    if (g->n_hits == g->cap_hits) {
        g->cap_hits = g->cap_hits ? g->cap_hits * 2 : 16;
        g->hits = (ffo_hit *)realloc(g->hits, sizeof(ffo_hit) * (size_t)g->cap_hits);
    }
    g->hits[g->n_hits].target = target;
    g->hits[g->n_hits].positions = positions;
    g->hits[g->n_hits].n_pos = npos;
    g->n_hits++;
    g->current_total += npos; /* getOffTargetCount = coordinates.size, crispr/CRISPRHit.scala:42 */
}

static void agg_update_ot(ffo_agg *a, int gi, uint64_t target, const uint64_t *positions, int npos) { /* ResultsAggregator.scala:61-69 */
    ffo_guide_ot *g = &a->guides[gi];
    if (!site_full(g)) {
        site_add_ot(g, target, positions, npos);
        if (site_full(g) && a->overflow_cb) a->overflow_cb(a->cb_ctx, gi);
    }
}

/* ------------------------------------------------------------------------------------------------
 * block compare -- reference/binary/blocks/BlockManager.scala:63-90, 143-254
 * `guides` is an array of aggregator indices (GuideIndex.index); guide longs come from the aggregator.
 * ---------------------------------------------------------------------------------------------- */
static int compare_linear_block(const ffo_pack *p, const int64_t *blk, size_t n, const int *guides, int n_guides,
                                ffo_agg *agg, int max_mm) { /* :212-254 */
    size_t off = 0;
    while (off < n) {                                    /* :225 */
        uint64_t target = (uint64_t)blk[off];            /* :231 */
        int count = ffo_get_count(target);               /* :232 */
        if (count <= 0) { ffo_set_error("Encoded position count should be greater than zero"); return -1; } /* :234 */
        if (n < off + (size_t)count) { ffo_set_error("Failed to correctly parse block"); return -2; }      /* :236 */
        const uint64_t *positions = (const uint64_t *)&blk[off + 1]; /* :239 (slice; we keep a view) */
        for (int gi = 0; gi < n_guides; gi++) {          /* :244 */
            g_all_comparisons++;                         /* :245 */
            int mm = ffo_mismatches(p, agg->guides[guides[gi]].encoding, target, FFO_STRING_MASK); /* :246 */
            if (mm <= max_mm) agg_update_ot(agg, guides[gi], target, positions, count);          /* :247-249 */
        }
        off += (size_t)count + 1;                        /* :252 */
    }
    return 0;
}

static int compare_indexed_block(const ffo_pack *p, const int64_t *blk, size_t n, const int *guides, int n_guides,
                                 ffo_agg *agg, int max_mm, const ffo_bin_and_mask *parent,
                                 const ffo_bin_and_mask *sub /* 256 */, int nsub, int *scratch) { /* :143-201 */
    int last_pos = 0, last_size = 0;
    for (int bi = 0; bi < nsub; bi++) {                  /* :160 */
        int64_t ps = blk[bi];
        int pos = (int)(ps >> 32);                       /* :164 */
        int size = (int)((int64_t)((uint64_t)ps << 32) >> 32); /* :165 */
        if (last_pos != 0 && pos >= 0 && pos != last_pos + last_size) { /* :167-168 */
            ffo_set_error("indexed block: last position %d plus size %d != pos %d", last_pos, last_size, pos);
            return -3;
        }
        last_pos = pos > 0 ? pos : 0;                    /* :170 */
        last_size = size;
        if (pos >= 0 && size > 0) {                      /* :174 */
            if ((size_t)nsub + (size_t)pos + (size_t)size > n) { ffo_set_error("indexed block: slice out of range"); return -4; }
            uint64_t full_mask = sub[bi].guide_mask | parent->guide_mask; /* :179 */
            uint64_t full_bin = parent->bin_long | sub[bi].bin_long;
            int m = 0;
            for (int gi = 0; gi < n_guides; gi++)        /* :187-191 */
                if (ffo_mismatches(p, agg->guides[guides[gi]].encoding, full_bin, full_mask) <= max_mm) scratch[m++] = guides[gi];
            if (m > 0) {                                 /* :194-196 */
                int rc = compare_linear_block(p, blk + nsub + pos, (size_t)size, scratch, m, agg, max_mm);
                if (rc) return rc;
            }
        }
    }
    return 0;
}

typedef struct block_manager { /* class BlockManager(offset, width=4, bitEncoding) :40-49 */
    const ffo_pack *p;
    int nsub;
    ffo_bin_and_mask *sub;
    int *scratch;
} block_manager;

static void block_manager_init(block_manager *bm, const ffo_pack *p, int offset, int width, int max_guides) {
    bm->p = p;
    bm->nsub = 1 << (2 * width);
    bm->sub = (ffo_bin_and_mask *)malloc(sizeof(ffo_bin_and_mask) * (size_t)bm->nsub);
    char name[16];
    for (int i = 0; i < bm->nsub; i++) { /* blockDescriptorLookup :46-49: binToLongComparitor(bin, offset) */
        ffo_bin_name(width, (uint32_t)i, name);
        ffo_bin_to_long_comparitor(p, name, width, offset, &bm->sub[i]);
    }
    bm->scratch = (int *)malloc(sizeof(int) * (size_t)(max_guides > 0 ? max_guides : 1));
}
static void block_manager_free(block_manager *bm) { free(bm->sub); free(bm->scratch); }

static int compare_block(block_manager *bm, const int64_t *blk, size_t n, const int *guides, int n_guides,
                         ffo_agg *agg, int max_mm, const ffo_bin_and_mask *bin) { /* :63-90 */
    if (n == 0) { ffo_set_error("empty block"); return -10; }
    int64_t first = blk[0]; /* :72 */
    if (first == 1) return compare_linear_block(bm->p, blk + 1, n - 1, guides, n_guides, agg, max_mm);        /* :75-79 */
    if (first == 2) {                                                                                         /* :80-84 */
        if (n - 1 < (size_t)bm->nsub) { ffo_set_error("indexed block shorter than its table"); return -11; }
        return compare_indexed_block(bm->p, blk + 1, n - 1, guides, n_guides, agg, max_mm, bin, bm->sub, bm->nsub, bm->scratch);
    }
    ffo_set_error("Invalid bin type, unknown value: %lld", (long long)first); /* :85-87 */
    return -12;
}

/* ------------------------------------------------------------------------------------------------
 * traversal -- reference/traversal/OrderedBinTraversalFactory.scala, LinearTraversal.scala
 * ---------------------------------------------------------------------------------------------- */
typedef struct linear_trav { /* LinearTraversal.scala:32-99 */
    int *guides_to_use;
    int n_use;
} linear_trav;

static void linear_overflow(void *ctx, int gi) { /* overflowGuide :64-76 */
    linear_trav *t = (linear_trav *)ctx;
    int w = 0;
    for (int i = 0; i < t->n_use; i++)
        if (t->guides_to_use[i] != gi) t->guides_to_use[w++] = t->guides_to_use[i];
    t->n_use = w;
}

typedef struct seek_trav { /* PrivateBinIterator, OrderedBinTraversalFactory.scala:72-129 */
    int *excluded;
    int n_excl, cap_excl;
} seek_trav;

static void seek_overflow(void *ctx, int gi) { /* :92 */
    seek_trav *t = (seek_trav *)ctx;
    if (t->n_excl == t->cap_excl) {
        t->cap_excl = t->cap_excl ? t->cap_excl * 2 : 16;
        t->excluded = (int *)realloc(t->excluded, sizeof(int) * (size_t)t->cap_excl);
    }
    t->excluded[t->n_excl++] = gi;
}

ffo_result *ffo_discover(const ffo_db *db, const uint64_t *guides, int n_guides, int max_mm, int max_ot,
                         int force_linear) { /* modules/OffTargetDiscovery.scala:98-135 */
    const ffo_pack *p = db->pack;
    ffo_result *res = (ffo_result *)calloc(1, sizeof *res);
    res->n = n_guides;
    res->guides = (ffo_guide_ot *)calloc((size_t)(n_guides > 0 ? n_guides : 1), sizeof(ffo_guide_ot));
    for (int i = 0; i < n_guides; i++) { /* new CRISPRSiteOT(guide, encoding, maximumOffTargets) :100-102 */
        res->guides[i].encoding = guides[i];
        res->guides[i].overflow = max_ot;
    }
    ffo_agg agg = {res->guides, n_guides, NULL, NULL};

    int nb = db->n_bins, w = db->bin_width;
    ffo_bin_and_mask *bins = (ffo_bin_and_mask *)malloc(sizeof(ffo_bin_and_mask) * (size_t)nb);
    char name[16];
    for (int b = 0; b < nb; b++) { /* binArray, OrderedBinTraversalFactory.scala:56 */
        ffo_bin_name(w, (uint32_t)b, name);
        ffo_bin_to_long_comparitor(p, name, w, 0, &bins[b]);
    }

    /* ---- OrderedBinTraversalFactory constructor :137-183 ---- */
    int saturated = 0;
    int **bin_guides = NULL;
    int *bin_nguides = NULL;
    if (!force_linear) {
        bin_guides = (int **)calloc((size_t)nb, sizeof(int *));
        bin_nguides = (int *)calloc((size_t)nb, sizeof(int));
        int needed = 0;
        int *tmp = (int *)malloc(sizeof(int) * (size_t)(n_guides > 0 ? n_guides : 1));
        for (int index = 0; index < nb; index++) {            /* :146 */
            int m = 0;
            for (int gi = 0; gi < n_guides; gi++)             /* :151-156 */
                if (ffo_mismatch_bin(p, &bins[index], guides[gi]) <= max_mm) tmp[m++] = gi;
            if (m > 0) {                                      /* :158-159 */
                bin_guides[index] = (int *)malloc(sizeof(int) * (size_t)m);
                memcpy(bin_guides[index], tmp, sizeof(int) * (size_t)m);
                bin_nguides[index] = m;
                needed++;
            }
            if (index % 500 == 0) {                           /* :161 statusInterval */
                double sat = (double)needed / (index > 0 ? (double)(index + 1) : 1.0); /* :162 */
                if (sat >= 0.95 && index >= 500) {            /* :165 */
                    saturated = 1;                            /* :168 */
                    break;                                    /* index = binArray.size :167 */
                }
            }
        }
        free(tmp);
        if ((double)needed / (double)nb >= 0.95) saturated = 1; /* :175-177 */
    }
    res->saturated = (force_linear || saturated);

    block_manager bm;
    block_manager_init(&bm, p, w, 4, n_guides); /* new BlockManager(header.binWidth, 4, bitCoder), SeekTraverser.scala:71 */
    int rc = 0;

    if (force_linear || saturated) { /* LinearTraversal + LinearTraverser.scan, OffTargetDiscovery.scala:120-125 */
        linear_trav lt;
        lt.guides_to_use = (int *)malloc(sizeof(int) * (size_t)(n_guides > 0 ? n_guides : 1));
        lt.n_use = n_guides;
        for (int i = 0; i < n_guides; i++) lt.guides_to_use[i] = i;
        agg.overflow_cb = linear_overflow;
        agg.cb_ctx = &lt;
        int *run = (int *)malloc(sizeof(int) * (size_t)(n_guides > 0 ? n_guides : 1));
        for (int b = 0; b < nb && rc == 0; b++) { /* every bin, in generator order: LinearTraversal.next :82-97 */
            int m = 0;
            for (int i = 0; i < lt.n_use; i++)
                if (ffo_mismatch_bin(p, &bins[b], guides[lt.guides_to_use[i]]) <= max_mm) run[m++] = lt.guides_to_use[i];
            const ffo_bin *bin = &db->bins[b];
            if (!bin->longs) { ffo_set_error("bin %d missing from database", b); rc = -20; break; }
            rc = compare_block(&bm, bin->longs, bin->n_longs, run, m, &agg, max_mm, &bins[b]); /* LinearTraverser.scala:94-100 */
        }
        free(run);
        free(lt.guides_to_use);
    } else { /* factory.iterator + SeekTraverser.scan, OffTargetDiscovery.scala:126-131 */
        seek_trav st = {NULL, 0, 0};
        agg.overflow_cb = seek_overflow;
        agg.cb_ctx = &st;
        /* PrivateBinIterator: the first needed bin is cached unfiltered (:81-87); each next() caches the following
         * needed bin filtered by the guides excluded SO FAR, i.e. before the returned bin is processed (:107-118). */
        int cached = -1, *cached_list = NULL, cached_n = 0, scan = 0;
        for (; scan < nb && cached < 0; scan++)
            if (bin_nguides[scan] > 0) {
                cached = scan;
                cached_n = bin_nguides[scan];
                cached_list = (int *)malloc(sizeof(int) * (size_t)cached_n);
                memcpy(cached_list, bin_guides[scan], sizeof(int) * (size_t)cached_n);
            }
        while (cached >= 0 && rc == 0) {
            int cur = cached, *cur_list = cached_list, cur_n = cached_n;
            cached = -1; cached_list = NULL; cached_n = 0;
            for (; scan < nb && cached < 0; scan++)
                if (bin_nguides[scan] > 0) {
                    cached = scan;
                    cached_list = (int *)malloc(sizeof(int) * (size_t)bin_nguides[scan]);
                    for (int i = 0; i < bin_nguides[scan]; i++) { /* filter(!guidesToExclude.contains) :109 */
                        int g = bin_guides[scan][i], ex = 0;
                        for (int k = 0; k < st.n_excl; k++) if (st.excluded[k] == g) { ex = 1; break; }
                        if (!ex) cached_list[cached_n++] = g;
                    }
                }
            const ffo_bin *bin = &db->bins[cur];
            if (!bin->longs) { ffo_set_error("bin %d missing from database", cur); rc = -20; }
            else rc = compare_block(&bm, bin->longs, bin->n_longs, cur_list, cur_n, &agg, max_mm, &bins[cur]); /* SeekTraverser.scala:84-90 */
            free(cur_list);
        }
        free(cached_list);
        free(st.excluded);
    }
    block_manager_free(&bm);
    if (bin_guides) { for (int b = 0; b < nb; b++) free(bin_guides[b]); free(bin_guides); }
    free(bin_nguides);
    free(bins);
    if (rc) { ffo_result_free(res); return NULL; }
    return res;
}

/* The linear traversal (LinearTraversal.next :82-97 + LinearTraverser.scan :94-100) over the bins [bin_begin, bin_end) only: what one
 * worker of a run split over the bins does (SURVEY.md section 8d asks for the CPU port on all host cores: bins are independent, a
 * worker filters and scans its own range).  The cut-off applies within the range; a throughput baseline, not a result path. */
ffo_result *ffo_discover_bin_range(const ffo_db *db, const uint64_t *guides, int n_guides, int max_mm, int max_ot, int bin_begin, int bin_end) {
    const ffo_pack *p = db->pack;
    int nb = db->n_bins, w = db->bin_width;
    if (bin_begin < 0) bin_begin = 0;
    if (bin_end > nb) bin_end = nb;
    ffo_result *res = (ffo_result *)calloc(1, sizeof *res);
    res->n = n_guides;
    res->saturated = 1;
    res->guides = (ffo_guide_ot *)calloc((size_t)(n_guides > 0 ? n_guides : 1), sizeof(ffo_guide_ot));
    for (int i = 0; i < n_guides; i++) { res->guides[i].encoding = guides[i]; res->guides[i].overflow = max_ot; }
    ffo_agg agg = {res->guides, n_guides, NULL, NULL};
    block_manager bm;
    block_manager_init(&bm, p, w, 4, n_guides);
    linear_trav lt;
    lt.guides_to_use = (int *)malloc(sizeof(int) * (size_t)(n_guides > 0 ? n_guides : 1));
    lt.n_use = n_guides;
    for (int i = 0; i < n_guides; i++) lt.guides_to_use[i] = i;
    agg.overflow_cb = linear_overflow;
    agg.cb_ctx = &lt;
    int *run = (int *)malloc(sizeof(int) * (size_t)(n_guides > 0 ? n_guides : 1));
    int rc = 0;
    char name[16];
    for (int b = bin_begin; b < bin_end && rc == 0; b++) {
        ffo_bin_and_mask bam;
        ffo_bin_name(w, (uint32_t)b, name);
        ffo_bin_to_long_comparitor(p, name, w, 0, &bam);
        int m = 0;
        for (int i = 0; i < lt.n_use; i++)
            if (ffo_mismatch_bin(p, &bam, guides[lt.guides_to_use[i]]) <= max_mm) run[m++] = lt.guides_to_use[i];
        const ffo_bin *bin = &db->bins[b];
        if (!bin->longs) { ffo_set_error("bin %d missing from database", b); rc = -20; break; }
        rc = compare_block(&bm, bin->longs, bin->n_longs, run, m, &agg, max_mm, &bam);
    }
    free(run);
    free(lt.guides_to_use);
    block_manager_free(&bm);
    if (rc) { ffo_result_free(res); return NULL; }
    return res;
}

void ffo_result_free(ffo_result *r) {
    if (!r) return;
    for (int i = 0; i < r->n; i++) {
        free(r->guides[i].hits);
        free(r->guides[i].hit_cfd);
        free(r->guides[i].contig); free(r->guides[i].bases); free(r->guides[i].context);
        for (int k = 0; k < r->guides[i].n_owned; k++) free(r->guides[i].owned[k]);
        free(r->guides[i].owned);
    }
    free(r->guides);
    free(r);
}
int ffo_result_n_guides(const ffo_result *r) { return r->n; }
int ffo_result_saturated(const ffo_result *r) { return r->saturated; }
int ffo_result_n_hits(const ffo_result *r, int g) { return r->guides[g].n_hits; }
int ffo_result_current_total(const ffo_result *r, int g) { return r->guides[g].current_total; }
int ffo_result_full(const ffo_result *r, int g) { return site_full(&r->guides[g]); }
uint64_t ffo_result_hit_target(const ffo_result *r, int g, int h) { return r->guides[g].hits[h].target; }
int ffo_result_hit_npos(const ffo_result *r, int g, int h) { return r->guides[g].hits[h].n_pos; }
const uint64_t *ffo_result_hit_positions(const ffo_result *r, int g, int h) { return r->guides[g].hits[h].positions; }

size_t ffo_result_total_positions(const ffo_result *r) {
    size_t t = 0;
    for (int g = 0; g < r->n; g++)
        for (int h = 0; h < r->guides[g].n_hits; h++) t += (size_t)r->guides[g].hits[h].n_pos;
    return t;
}

size_t ffo_result_export(const ffo_result *r, uint64_t *goff, uint64_t *targets, uint64_t *poff, uint64_t *positions) {
    size_t H = 0, P = 0;
    for (int g = 0; g < r->n; g++) {
        if (goff) goff[g] = H;
        for (int h = 0; h < r->guides[g].n_hits; h++) {
            const ffo_hit *hit = &r->guides[g].hits[h];
            if (targets) targets[H] = hit->target;
            if (poff) poff[H] = P;
            if (positions) memcpy(positions + P, hit->positions, sizeof(uint64_t) * (size_t)hit->n_pos);
            P += (size_t)hit->n_pos;
            H++;
        }
    }
    if (goff) goff[r->n] = H;
    if (poff) poff[H] = P;
    return H;
}
