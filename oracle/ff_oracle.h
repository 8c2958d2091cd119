/*
 * ff_oracle.h -- CPU ORACLE for the FlashFry `discover` + off-target scoring path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain-C restatement of the reference
 * algorithm (mckennalab/FlashFry v1.15, Scala) kept in the reference's own loop structure so that
 * (a) the HIP path can be checked bit-for-bit against it and (b) it can be timed as the CPU baseline.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under
 * flashfry_amd/ links, imports or executes anything from this directory.
 *
 * Pinning: the reference is Scala/JVM and cannot be built or run in this environment (no JVM), so the
 * oracle is pinned by the golden vectors of the reference's own unit tests (tests/test_oracle_golden.py):
 * BitEncodingTest, BitPositionTest, UtilsTest, Doench2016CFDScoreTest, CrisprMitEduOffTargetTest,
 * ClosestHitTest, SimpleSiteFinderTest, TabDelimitedHanderTest (+ fixtures fake.sites and
 * test_blockAACCTTGG.binary).  The BGZF container codec lives in htsjdk 2.8.1 (not in the reference
 * tree; build.sbt:18) and no reference test exercises it: parity is UNPINNED at that one boundary
 * (public format: SAM spec 4.1; cross-checked against Python's gzip module instead).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/src/main/scala unless noted).
 */
#ifndef FF_ORACLE_H
#define FF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enzyme parameter packs: standards/StandardScanParameters.scala:50-215 ---- */
typedef struct ffo_pack {
    int index;            /* ParameterPack.parameterPackToIndex, :72-80 (1=Cpf1 2=Cas9 3=NGG 4=NAG 5=Cas9-19 6=NGG-19) */
    const char *name;
    int scan_len;         /* totalScanLength */
    int pam_len;          /* pamLength */
    int five_prime;       /* fivePrimePam */
    uint64_t cmp_mask;    /* comparisonBitEncoding */
    int guide_lo, guide_hi; /* guideRange */
    int cas9_23;          /* enzymeParent == Cas9Type && scan_len == 23 (validity of CFD / Hsu2013) */
} ffo_pack;

const ffo_pack *ffo_pack_by_index(int index);
const ffo_pack *ffo_pack_by_name(const char *name);

/* ---- bit codecs: bitcoding/BitEncoding.scala ---- */
#define FFO_STRING_MASK 0xFFFFFFFFFFFFULL
#define FFO_UPPER_BITS  0xAAAAAAAAAAAAULL

int      ffo_bit_encode(const char *s, int len, int count, uint64_t *out);          /* :46-67 ; 0 ok, <0 error */
int      ffo_bit_decode(uint64_t enc, int actual_size, char *out_str);              /* :85-99 ; returns count (signed short) */
int      ffo_get_count(uint64_t enc);                                               /* :114 */
uint64_t ffo_update_count(uint64_t enc, int count);                                 /* :108-111 */
int      ffo_mismatches(const ffo_pack *p, uint64_t e1, uint64_t e2, uint64_t additional_mask); /* :127-132 */

typedef struct ffo_bin_and_mask { uint64_t bin_long; uint64_t guide_mask; } ffo_bin_and_mask;
uint64_t ffo_bin_shift(const ffo_pack *p, int bin_size, uint64_t base, int right_shift_bases);      /* :179-185 */
uint64_t ffo_comp_bitmask_for_bin(const ffo_pack *p, int bin_size, int right_shift_bases);          /* :167-170 */
int      ffo_bin_to_long_comparitor(const ffo_pack *p, const char *bin, int bin_size,
                                    int right_shift_bases, ffo_bin_and_mask *out);                   /* :153-157 */
int      ffo_mismatch_bin(const ffo_pack *p, const ffo_bin_and_mask *bin, uint64_t guide);          /* :142-144 */

/* global counters mirroring BitEncoding.allComparisons (:193) and Traverser.allComparisons (Traverser.scala:74) */
uint64_t ffo_counter_bit_comparisons(void);
uint64_t ffo_counter_all_comparisons(void);
void     ffo_counters_reset(void);

/* bins: utils/BaseCombinationGenerator.scala:24-69 (A<C<G<T lexicographic == numeric order) */
void ffo_bin_name(int width, uint32_t bin_index, char *out /* width+1 */);

/* byte <-> long, native (little-endian) order: utils/Utils.scala:154-186 */
void ffo_longs_to_bytes(const int64_t *longs, size_t n, uint8_t *out);
void ffo_bytes_to_longs(const uint8_t *bytes, size_t nbytes, int64_t *out);

/* ---- positions: bitcoding/BitPosition.scala:51-92 ---- */
uint64_t ffo_pos_encode(int contig_id, uint32_t position, int target_len, int forward);
void     ffo_pos_decode(uint64_t enc, int *contig_id, uint32_t *start, int *size, int *forward);

/* ---- database blocks: reference/binary/blocks/BlockManager.scala ---- */
/* targets[] carry their count in bits 63:48; positions are concatenated, count(target i) longs each. */
size_t ffo_create_linear_block(const uint64_t *targets, const uint64_t *positions, size_t n_targets,
                               int64_t *out /* NULL = size query */);                               /* :424-442 */
size_t ffo_create_indexed_block(const ffo_pack *p, const uint64_t *targets, const uint64_t *positions,
                                size_t n_targets, int prefix_len, int lookup_bin_size,
                                int64_t *out /* NULL = size query */);                              /* :362-413 */

/* ---- in-memory database = header + decoded bins ---- */
typedef struct ffo_db ffo_db;
ffo_db *ffo_db_new(int enzyme_index, int bin_width);
void    ffo_db_free(ffo_db *db);
/* copies the block; longs[0] is the block type (1 linear, 2 indexed) */
int     ffo_db_set_bin(ffo_db *db, uint32_t bin_index, const int64_t *longs, size_t n_longs, int n_targets);
int     ffo_db_add_contig(ffo_db *db, const char *name);
int     ffo_db_n_bins(const ffo_db *db);
/* test aid (tools/stress_parity.py --seal): move every bin's longs and the bin table into ONE mapping and make it read-only, so that a
 * stray CPU store into the checker's database faults where it happens instead of changing an answer; 0 on success.  The database
 * cannot be modified afterwards (ffo_db_set_bin fails). */
int     ffo_db_seal(ffo_db *db);
uint64_t ffo_db_checksum(const ffo_db *db, uint64_t *per_bin, const uint64_t *expect, int *first_changed);   /* test aid: see ff_oracle.c */
int     ffo_db_bin_width(const ffo_db *db);
int     ffo_db_enzyme(const ffo_db *db);
int     ffo_db_n_contigs(const ffo_db *db);
const char *ffo_db_contig(const ffo_db *db, int id /* 1-based */);
size_t  ffo_db_bin_longs(const ffo_db *db, uint32_t bin_index, const int64_t **longs, int *n_targets);
/* Build every bin from a globally sorted, de-duplicated target list the way DatabaseWriter does:
 * linear if <= max_linear targets or 5'-PAM enzyme, else indexed (DatabaseWriter.scala:85-89). */
int     ffo_db_build_from_sorted(ffo_db *db, const uint64_t *targets, const uint64_t *positions,
                                 size_t n_targets, int max_targets_per_linear_bin);
/* on-disk format: text .header (BinaryHeader.scala:69-160) + BGZF body (DatabaseWriter.scala:58-111) */
int     ffo_db_write(const ffo_db *db, const char *path);
ffo_db *ffo_db_read(const char *path);
const char *ffo_last_error(void);

/* ---- discover: modules/OffTargetDiscovery.scala:79-153 over an in-memory database ---- */
typedef struct ffo_result ffo_result;
/* guides[] = bitEncodeString(StringCount(bases,1)); results are reported in input order (the aggregator's
 * sort by start, ResultsAggregator.scala:35, only changes output row order -- applied by ffo_discover_fasta) */
ffo_result *ffo_discover(const ffo_db *db, const uint64_t *guides, int n_guides,
                         int max_mismatch, int max_offtargets, int force_linear);
/* one worker of a run split over the bins: the linear traversal over [bin_begin, bin_end) only (CPU baseline on all host cores) */
ffo_result *ffo_discover_bin_range(const ffo_db *db, const uint64_t *guides, int n_guides, int max_mismatch, int max_offtargets,
                                   int bin_begin, int bin_end);
void        ffo_result_free(ffo_result *r);
int         ffo_result_n_guides(const ffo_result *r);
int         ffo_result_saturated(const ffo_result *r);      /* traversal mode actually used: 1 linear, 0 seek */
int         ffo_result_n_hits(const ffo_result *r, int guide);
int         ffo_result_current_total(const ffo_result *r, int guide);
int         ffo_result_full(const ffo_result *r, int guide);
uint64_t    ffo_result_hit_target(const ffo_result *r, int guide, int hit);
int         ffo_result_hit_npos(const ffo_result *r, int guide, int hit);
const uint64_t *ffo_result_hit_positions(const ffo_result *r, int guide, int hit);
/* bulk export: CSR, returns total hits; arrays may be NULL */
size_t      ffo_result_export(const ffo_result *r, uint64_t *guide_offsets /* n+1 */, uint64_t *hit_targets,
                              uint64_t *pos_offsets /* H+1 */, uint64_t *positions);
size_t      ffo_result_total_positions(const ffo_result *r);

/* ---- scoring over a guide's retained hit list ---- */
double ffo_cfd_score_pair(const char *guide20, const char *ot20);                       /* Doench2016CFDScore.scala:132-151 */
double ffo_cfd_pam(const char *pam2);                                                   /* :211-214 */
double ffo_jost_calc_score(const ffo_pack *p, const char *target, const char *off_target);  /* JostAndSantosCRISPRi.scala:92-127; NaN where it throws */
double ffo_hsu_score_offtarget(const ffo_pack *p, const char *guide_bases, uint64_t ot);/* CrisprMitEduOffTarget.scala:107-148 */

typedef struct ffo_guide_scores {
    /* Doench2016CFDScore.scoreGuide :53-88 */
    double cfd_max;            /* raw max (before the 0.023 threshold) */
    double cfd_spec;           /* specificity score */
    int    cfd_valid;
    /* CrisprMitEduOffTarget.score_crispr :60-105 */
    double hsu;
    int    hsu_valid;
    /* ClosestHit.scoreGuide :43-76 */
    int    closest;            /* INT_MAX -> "UNK" */
    int    closest_count;
    int    hist[5];
    /* DangerousSequences :61-65 */
    int    in_genome;
    /* JostAndSantosCRISPRi.scoreGuide :27-46 (valid for Cas9 20-mers and 19-mers, :53-58) */
    int    jost_valid;
    double jost_max;           /* 0.0 when no hit was scored (:43) */
    double jost_spec;
} ffo_guide_scores;
/* per_hit_cfd (optional, n_hits doubles): pam*cfd for scored hits, NaN for hits skipped as on-target */
int ffo_score_guide(const ffo_pack *p, uint64_t guide, const uint64_t *hit_targets, int n_hits,
                    ffo_guide_scores *out, double *per_hit_cfd);
int ffo_result_score_guide(const ffo_result *r, const ffo_db *db, int guide, ffo_guide_scores *out, double *per_hit_cfd);

/* ---- config C5 (no reference counterpart; specification: DESIGN.md section 8 "f4"): best alignment of a Cas12a guide with a
 * target allowing one bulge of one base.  Returns the mismatch count of the best alignment (fewest mismatches; ties none <
 * RNA bulge < DNA bulge, then the smallest position); *type = 0 none / 1 RNA / 2 DNA, *pos = bulge position k (1..18) or 0. */
int ffo_bulge_align(const ffo_pack *p, uint64_t guide, uint64_t target, int max_bulge, int *type, int *pos);

/* java.lang.Double.toString semantics (shortest repr that round-trips; sci notation <1e-3 or >=1e7) */
int ffo_java_double_to_string(double d, char *out /* >= 32 bytes */);

/* ---- guide discovery in FASTA text: reference/ReferenceEncoder.scala:104-175 ---- */
typedef struct ffo_site { int start; int forward; char bases[25]; char context[64]; int has_context; } ffo_site;
/* scans ONE contig's (upper-cased, concatenated) sequence; returns number of sites, fills up to cap */
int ffo_find_sites(const ffo_pack *p, const char *seq, size_t len, int flank, ffo_site *out, int cap);

/* ---- end-to-end file drivers (reference CLI restated): return 0 on success ---- */
int ffo_index_fasta(const char *fasta_path, const char *db_path, const char *enzyme_name, int bin_width);   /* modules/BuildOffTargetDatabase.scala:57-89 */
int ffo_discover_fasta(const char *db_path, const char *fasta_path, const char *out_path, int max_mismatch,
                       int max_offtargets, int flank, int position_output, int force_linear,
                       double min_gc, double max_gc);                                                        /* modules/OffTargetDiscovery.scala:79-153 */
int ffo_score_file(const char *db_path, const char *in_path, const char *out_path, const char *metrics_csv,
                   int max_mismatch, int include_ots, int max_reciprocal_mismatch);                                                       /* modules/ScoreResults.scala:90-154 */

#ifdef __cplusplus
}
#endif
#endif
