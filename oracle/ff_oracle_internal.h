/* ff_oracle_internal.h -- shared structs of the CPU oracle (TEST INFRASTRUCTURE; see ff_oracle.h). */
#ifndef FF_ORACLE_INTERNAL_H
#define FF_ORACLE_INTERNAL_H

#include "ff_oracle.h"

typedef struct ffo_bin {
    int64_t *longs;    /* decoded block, longs[0] = block type */
    size_t n_longs;
    int n_targets;
} ffo_bin;

struct ffo_db {
    const ffo_pack *pack;
    int bin_width;
    int n_bins;
    ffo_bin *bins;
    char **contigs; /* id = index + 1, bitcoding/BitPosition.scala:38-49 */
    int n_contigs;
    void *sealed;        /* ffo_db_seal: every bin's longs AND the bin table live in this one read-only mapping */
    size_t sealed_bytes;
};

typedef struct ffo_hit { /* crispr/CRISPRHit.scala:39-43 */
    uint64_t target;
    const uint64_t *positions;
    int n_pos;
} ffo_hit;

typedef struct ffo_guide_ot { /* crispr/CRISPRSiteOT.scala:31-39 (+ the CRISPRSite it wraps, crispr/CRISPRSite.scala:34-53) */
    uint64_t encoding;
    int overflow;
    int inherited_overflow;
    int current_total;
    ffo_hit *hits;
    int n_hits, cap_hits;
    double *hit_cfd;      /* per-hit Doench2016CFDScore annotation (NaN = none), CRISPRHit.addScore */
    /* site fields, only filled by the file drivers */
    char *contig, *bases, *context;
    int start, forward;
    int valid_coords;     /* hits re-read from a file without positions: validOffTargetCoordinates=false */
    uint64_t **owned;     /* position arrays owned by this guide (file reader) */
    int n_owned;
} ffo_guide_ot;

struct ffo_result {
    ffo_guide_ot *guides;
    int n;
    int saturated;
};

void ffo_set_error(const char *fmt, ...);
uint32_t ffo_target_bin(const ffo_pack *p, int bin_width, uint64_t target);

#endif
