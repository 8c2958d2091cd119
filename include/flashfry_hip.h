/*
 * flashfry_hip.h -- C ABI of the MI355X-native FlashFry `discover` scan + off-target score aggregation.
 *
 * This is the drop-in boundary.  It replaces, for ONE hot path, what the reference does behind
 *
 *   trait Traverser { def scan(binaryFile, header, traversal, aggregator, maxMismatch, configuration,
 *                              bitCoder, posCoder): Unit }
 *       src/main/scala/reference/traverser/Traverser.scala:38-61
 *       (implementations SeekTraverser.scala:58-121, LinearTraverser.scala:59-130; chosen in
 *        modules/OffTargetDiscovery.scala:119-135; hits are delivered through
 *        ResultsAggregator.updateOT, crispr/ResultsAggregator.scala:61-69)
 *
 * and, for the scoring epilogue, the hit-list models behind
 *
 *   trait ScoreModel / SingleGuideScoreModel.scoreGuide      scoring/ScoreModel.scala:31-133
 *       Doench2016CFDScore.scala:53-88,132-151   CrisprMitEduOffTarget.scala:60-148
 *       ClosestHit.scala:43-76                    DangerousSequences.scala:61-65
 *
 * A JVM host binds these entry points with a thin JNI stub (INTEGRATION.md shows it); this repository's own
 * C++ CLI and the ctypes binding used by the tests call exactly the same symbols.
 *
 * Conventions: plain C, no C++ or torch types; every function returns 0 on success or a negative FFH_E* code
 * (never throws); ffh_last_error() gives the message.  The caller owns every input buffer; the library owns
 * an ffh_result until ffh_result_free().  A context is bound to one GPU and one HIP stream and is NOT
 * thread-safe (the reference path is single-threaded and not re-entrant either, Traverser.scala:68-74); use
 * one context per host thread / per GPU.  All integers are host (little-endian) order.  Guide and target
 * longs use the reference's layout verbatim (bitcoding/BitEncoding.scala:46-67: 2 bits per base, first base
 * most significant, occurrence count in bits 63:48), so GuideIndex.guide values pass through unchanged.
 */
#ifndef FLASHFRY_HIP_H
#define FLASHFRY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFH_VERSION 1

enum {
    FFH_OK = 0,
    FFH_E_ARG = -1,       /* bad argument (the reference would fail a require/assert) */
    FFH_E_HIP = -2,       /* a HIP runtime call failed (message carries hipGetErrorString) */
    FFH_E_FORMAT = -3,    /* malformed database block / header (BlockManager.scala:85-87, BinaryHeader.scala:121-124) */
    FFH_E_STATE = -4,     /* call order: no database loaded, no scan to finalize, ... */
    FFH_E_IO = -5,        /* file could not be read */
    FFH_E_NOMEM = -6,
    FFH_E_NODEVICE = -7   /* no usable GPU: there is deliberately NO CPU fallback */
};

typedef struct ffh_ctx ffh_ctx;
typedef struct ffh_result ffh_result;

int ffh_version(void);
int ffh_device_count(void);
/* Diagnostics, no reference counterpart: with FFH_POOL_DEBUG=1 in the environment the page-locked result blocks carry canaries and
 * released blocks a poison pattern; this is the number of violations seen so far in the process (0 when the checks are off). */
unsigned long long ffh_debug_pool_errors(void);

/* enzyme_index as stored in the database header: 1 Cpf1, 2 spCas9, 3 spCas9-NGG, 4 spCas9-NAG, 5 spCas9 19-mer,
 * 6 spCas9-NGG 19-mer (ParameterPack.indexToParameterPack, standards/StandardScanParameters.scala:61-69).
 * enzyme_index 0 defers the choice to the header of the database opened later (BinaryHeader.scala:127).
 * Returns NULL on failure; ffh_last_error(NULL) then tells why (the context-less message is kept per calling thread). */
ffh_ctx *ffh_create(int device_id, int enzyme_index);
void ffh_destroy(ffh_ctx *ctx);
const char *ffh_last_error(const ffh_ctx *ctx);

/* ---------------------------------------------------------------------------------------------------------
 * Database residency.  Exactly one of the loaders is called once per context; the database then stays in HBM
 * (targets, positions, and the two bucketed scan images built from them) for any number of discover calls.
 * ------------------------------------------------------------------------------------------------------- */

/* Decoded bin payloads, i.e. the Array[Long] that fillBlock/byteArrayToLong hand to BlockManager.compareBlock
 * (SeekTraverser.scala:113-120, Utils.scala:167-186), concatenated in bin order.  bin_offsets has n_bins+1
 * entries (in longs).  Block types 1 (linear) and 2 (indexed, 256-entry sub-bin table) are accepted,
 * anything else is FFH_E_FORMAT (BlockManager.scala:63-90). */
int ffh_db_load_blocks(ffh_ctx *ctx, const int64_t *longs, const uint64_t *bin_offsets, uint32_t n_bins);

/* Structure-of-arrays form: targets[] in database order (count in bits 63:48), positions[] concatenated with
 * count(target i) entries each.  on_device != 0 means both pointers are device pointers on this context's GPU
 * (they are copied; the caller keeps ownership). */
int ffh_db_load_soa(ffh_ctx *ctx, const uint64_t *targets, uint64_t n_targets, const uint64_t *positions,
                    uint64_t n_positions, int on_device);

/* On-disk database written by `index` (text <path>.header, BinaryHeader.scala:69-160, + BGZF body,
 * DatabaseWriter.scala:58-111).  Only bins [bin_begin, bin_end) are loaded: the static shard of this GPU.
 * bin_end == 0 means "to the last bin". */
int ffh_db_open(ffh_ctx *ctx, const char *db_path, uint32_t bin_begin, uint32_t bin_end);

/* Header only (enzyme, bin table, contig names) -- what `score` needs (modules/ScoreResults.scala:91): no GPU memory
 * is touched and nothing is scanned. */
int ffh_db_open_header(ffh_ctx *ctx, const char *db_path);
/* uncompressed payload bytes of a bin as recorded in the header (BlockOffset.uncompressedSize); used to balance shards */
uint64_t ffh_db_bin_bytes(const ffh_ctx *ctx, uint32_t bin);

typedef struct ffh_db_info {
    uint64_t n_targets;      /* unique target sequences resident (this shard) */
    uint64_t n_positions;    /* genomic positions resident (this shard) */
    uint32_t n_bins;         /* bins in the database header (0 for ffh_db_load_soa) */
    uint32_t bin_begin, bin_end;
    int enzyme_index;
    int prefix_bases;        /* bucket widths of the two resident scan images */
    int suffix_bases;
    double prepare_ms;       /* device time spent building the scan images */
} ffh_db_info;
int ffh_db_info_get(const ffh_ctx *ctx, ffh_db_info *out);

/* Writes a database in the reference's on-disk format: BGZF body bin by bin + text "<db_path>.header"
 * (replaces DatabaseWriter.writeToBinnedFile reference/binary/DatabaseWriter.scala:58-111, BlockManager.createLinearBlock /
 * createIndexedBlock blocks/BlockManager.scala:362-442 and BinaryHeader.writeHeader binary/BinaryHeader.scala:69-97).
 * targets[]: unique target longs in sequence order with their occurrence count in bits 63:48 (what BlockReader.scala:138-159
 * produces); positions[]: their position longs, count per target, same order; contigs[]: names, id = index + 1.
 * Host memory in, files out; no GPU involved.  Errors are reported through ffh_last_error(NULL). */
int ffh_db_write(const char *db_path, int enzyme_index, int bin_width, const char *const *contigs, uint32_t n_contigs, const uint64_t *targets,
                 uint64_t n_targets, const uint64_t *positions, uint64_t n_positions);

/* ---------------------------------------------------------------------------------------------------------
 * index: build a database from a reference genome on the GPU.  The host hands over one contig at a time
 * (upper- or lower-case bases, anything that is not ACGT never matches); the device finds the target sites of both
 * strands (SimpleSiteFinder, reference/ReferenceEncoder.scala:104-175), sorts them by sequence keeping discovery order
 * inside equal sequences and merges duplicates into (sequence, count <= 32767, positions) (BinWriter.scala:58-100,
 * BlockReader.scala:54-159); ffh_indexer_finish writes the result with ffh_db_write.
 * Replaces modules/BuildOffTargetDatabase.scala:57-89.  Errors: ffh_indexer_last_error.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct ffh_indexer ffh_indexer;
typedef struct ffh_index_stats {
    uint64_t n_bases, n_sites, n_targets, n_positions;
    uint32_t n_contigs, reserved;
    double scan_ms;   /* copies of the contigs to the device + site discovery */
    double sort_ms;   /* sort, merge of duplicates, copy of the result to the host */
    double write_ms;  /* ffh_db_write: blocks, BGZF, header (host threads) */
} ffh_index_stats;
ffh_indexer *ffh_indexer_create(int device_id, int enzyme_index);
void ffh_indexer_destroy(ffh_indexer *ix);
const char *ffh_indexer_last_error(const ffh_indexer *ix);
int ffh_indexer_add_contig(ffh_indexer *ix, const char *name, const char *sequence, uint64_t length);
int ffh_indexer_finish(ffh_indexer *ix, const char *db_path, int bin_width, ffh_index_stats *stats);

/* CPUs this process may really use (hardware threads, affinity mask, cgroup CPU quota; FFH_LOAD_THREADS overrides), <= 128:
 * what the loader, the writer and the CLI's text formatting size their thread pools with. */
int ffh_host_threads(void);

/* Where the time of the last ffh_db_open / ffh_db_load_blocks went (milliseconds of host wall time, stages in order). */
typedef struct ffh_load_stats {
    double open_ms;            /* header parse, mmap, BGZF member directory */
    double inflate_ms;         /* BGZF members -> payload longs in device memory: the file is staged through page-locked
                                  buffers with overlapped copies and inflated on the device (FFH_INFLATE=host: inflated by
                                  the host threads before the copy); ffh_db_load_blocks: the host-to-device copy */
    double decode_ms;          /* bin payloads -> targets[] / positions[] on the device */
    double prepare_ms;         /* scan images (= ffh_db_info.prepare_ms) */
    uint64_t compressed_bytes; /* BGZF bytes inflated */
    uint64_t raw_bytes;        /* payload bytes produced and copied to the device */
    uint32_t threads;          /* host threads that staged (or inflated) */
    uint32_t reserved;
    double device_inflate_ms;  /* part of inflate_ms spent in the device inflate + CRC kernels (0 with FFH_INFLATE=host) */
    double alloc_ms;           /* host wall time the calling thread spent inside hipMalloc / hipFree during the load (round 6): part of the
                                  stages above, wherever a buffer had to grow -- GB-sized allocations right after another context's have been
                                  freed are where a load can lose a second without a kernel running */
} ffh_load_stats;
int ffh_db_load_stats(const ffh_ctx *ctx, ffh_load_stats *out);
/* contig names of the database header (1-based ids as in BitPosition.scala:38-49); NULL past the end */
const char *ffh_db_contig(const ffh_ctx *ctx, uint32_t contig_id);

/* Force the candidate-generation split (tests / tuning).  prefix_bases = width of the prefix bucket key,
 * prefix_radius = mismatches tolerated inside the prefix; the suffix pass covers the rest.  -1 = automatic. */
int ffh_set_plan(ffh_ctx *ctx, int prefix_bases, int prefix_radius);

/* ---------------------------------------------------------------------------------------------------------
 * discover = scan + finalize.  Multi-GPU callers run ffh_scan on every shard, exchange ffh_shard_totals, and
 * pass the totals of the lower-ranked shards to ffh_finalize so the ordered cut-off of
 * CRISPRSiteOT.addOT/full (crispr/CRISPRSiteOT.scala:39-46) is applied across shards in database order.
 * ------------------------------------------------------------------------------------------------------- */

/* Finds every (guide, target) with mismatches(guide, target) <= max_mismatch (BitEncoding.scala:127-132) in
 * this shard; hits stay on the device, sorted by (guide, database order).  `guides` (here and in ffh_discover) may point to host
 * or to device memory of the context's GPU: a guide set that already sits in HBM is not staged through the host again. */
int ffh_scan(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch);

/* ffh_scan for a caller that will keep at most max_offtargets positions per guide (maximumOffTargets): the shard is scanned slab by
 * slab in database order and a guide whose positions so far reach the limit is not scanned against the later slabs -- the
 * reference stops feeding an overflowed guide as well (crispr/ResultsAggregator.scala:61-69, LinearTraversal.scala:64-76).  The
 * retained hits, totals and aggregates are exactly those of ffh_scan for every ffh_finalize / ffh_shard_totals limit <= max_offtargets
 * (a larger one makes the library redo the scan unbounded first: the guide set is still resident); the raw hits collected per guide stay within a small multiple of the limit instead of growing with
 * the size of the guide's repeat family.  3'-PAM enzymes only (database order must follow the compared bases); elsewhere, and
 * when bounding is switched off, it is ffh_scan.  ffh_discover scans this way when bounding is on.
 * ffh_set_bounding: 0 = never, 1 = always, -1 (default) = switch itself on for the context once a scan has collected more than
 * 2048 raw hits per guide, or could not be finished unbounded (more than 2^32 raw hits: ffh_discover's retry).  Once on it STAYS on for
 * the context -- the later scans of that context, ffh_scan_bounded / ffh_finalize included, are bounded -- until ffh_set_bounding(ctx, 0)
 * or (ctx, -1) takes it back; on a sharded call every shard's context decides for itself.  ffh_get_bounding: 1 if the context's next scan
 * will be bounded, 0 if not. */
int ffh_scan_bounded(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets);
int ffh_set_bounding(ffh_ctx *ctx, int mode);
int ffh_get_bounding(const ffh_ctx *ctx);

/* per guide: sum of positions over ALL hits of this shard, saturated at `clamp` (pass max_offtargets) */
int ffh_shard_totals(ffh_ctx *ctx, uint32_t *totals /* n_guides */, uint32_t clamp);

#define FFH_FINALIZE_SUMMARIES_ONLY 1u /* do not copy hit lists / positions to the host */
#define FFH_FINALIZE_JOST 2u           /* also fill ffh_guide_summary.jost_max / jost_sum (Cas9 enzymes) */
#define FFH_FINALIZE_PRIOR_ON_DEVICE 4u /* prior_totals is a device pointer (multi-GPU exchange without a host round trip) */
#define FFH_FINALIZE_NO_POSITIONS 8u   /* hit lists without the position arrays: ffh_result_positions / _pos_offsets return NULL and
                                          nothing is gathered or copied for them.  What `discover` needs unless --positionOutput is
                                          given (modules/OffTargetDiscovery.scala:51-53): its table prints sequence_count_mismatches,
                                          and the count is in bits 63:48 of the hit's target long */
#define FFH_FINALIZE_NO_HIT_SCORES 16u /* hit lists without the per-hit pam*cfd array (ffh_result_hit_cfd returns NULL): the reference's
                                          discover delivers sequences, counts, mismatches and positions (crispr/CRISPRHit,
                                          ResultsAggregator.scala:61-69), the scores are aggregates (ffh_guide_summary).  8 of the 17
                                          bytes per hit that cross the link */
int ffh_finalize(ffh_ctx *ctx, const uint32_t *prior_totals /* NULL = first shard */, int max_offtargets,
                 unsigned flags, ffh_result **out);

/* ffh_scan_bounded + ffh_finalize in one call -- what Traverser.scan(...) binds (reference/traverser/Traverser.scala:53-60).
 * ONE scan holds fewer than 2^32 raw hits (its segment arithmetic is 32-bit).  A guide set that collects more (<= 5-6 mismatches on a
 * repeat-rich genome) is not refused: the call first bounds the scan (the automatic rule of ffh_set_bounding: bounding then stays on for the context), and if that is not enough halves the
 * guide set as often as needed and concatenates the parts' results -- guides are independent of each other everywhere on the path, the
 * reference is slow on such a set, not wrong (reference/binary/blocks/BlockManager.scala:212-254).  ffh_discover_sharded and
 * ffh_discover_bulge do the same; the two-step ffh_scan / ffh_finalize report FFH_E_ARG ("more than 2^32 raw hits") and leave the
 * split to their caller.  After a split call the context holds the scan of the LAST part only. */
int ffh_discover(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets,
                 unsigned flags, ffh_result **out);

/* ---------------------------------------------------------------------------------------------------------
 * Config C5 of BASELINE.json: Cas12a (Cpf1, enzyme index 1) off-targets with up to max_mismatch mismatches AND up to
 * max_bulge (0 or 1) bulges of one base.  The reference has no bulge search; the specification is DESIGN.md section 8
 * ("f4") and csrc/ffh_bulge.hpp: alignments none / RNA bulge at guide position k / DNA bulge at target position k
 * (1 <= k <= 18, position 0 next to the PAM), best = fewest mismatches, ties none < RNA < DNA then smallest k.
 * Candidates come from the two resident scan images (csrc/ffh_bulge.hpp: one side of the bulge is a plain comparison of
 * consecutive bases, i.e. a bucket key within max_mismatch of the guide's); every candidate is evaluated in full; no cut-off,
 * no scores (CFD / Hsu2013 are not defined for Cas12a).  Hits per guide in database order.  FFH_BULGE_PAM_TTTV keeps only targets whose fourth PAM base is
 * not T (TTTV sites inside a TTTN database).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct ffh_bulge_result ffh_bulge_result;
#define FFH_BULGE_PAM_TTTV 1u
#define FFH_BULGE_BRUTE_FORCE 2u /* every guide against every target instead of the seeded candidate search (its checker) */
#define FFH_BULGE_NONE 0
#define FFH_BULGE_RNA 1
#define FFH_BULGE_DNA 2
int ffh_discover_bulge(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_bulge, unsigned flags,
                       ffh_bulge_result **out);
uint32_t ffh_bulge_result_n_guides(const ffh_bulge_result *r);
uint64_t ffh_bulge_result_n_hits(const ffh_bulge_result *r);
const uint64_t *ffh_bulge_result_guide_offsets(const ffh_bulge_result *r);      /* n_guides + 1 */
const uint64_t *ffh_bulge_result_hit_targets(const ffh_bulge_result *r);
const uint8_t *ffh_bulge_result_hit_mismatches(const ffh_bulge_result *r);
const uint8_t *ffh_bulge_result_hit_bulge_type(const ffh_bulge_result *r);      /* FFH_BULGE_* */
const uint8_t *ffh_bulge_result_hit_bulge_position(const ffh_bulge_result *r);  /* k, 0 for FFH_BULGE_NONE */
void ffh_bulge_result_free(ffh_bulge_result *r);

/* Device-resident halves of the multi-GPU exchange (one process per GPU, RCCL): the shard totals are written to, and the
 * summaries of the last ffh_finalize copied to, device buffers of the caller (n_guides x uint32 / n_guides x
 * sizeof(ffh_guide_summary) bytes), so that the collectives run on device memory.  Both return after the copy completed. */
int ffh_shard_totals_device(ffh_ctx *ctx, uint32_t *device_totals /* n_guides */, uint32_t clamp);
int ffh_summaries_to_device(ffh_ctx *ctx, void *device_summaries);
/* The reduction of the aggregates over the shards is three collectives on device buffers of the caller:
 *   ffh_exchange_pack   summaries -> max[n][4] f64 (overflow, cfd_max, jost_max, -closest), sum[n][10] i32 (n_hits, ot_count, hist[5],
 *                       in_genome, n_scored, 0), fsum[n][3] f64 (cfd_sum, hsu_sum, jost_sum);            then all-reduce MAX of max
 *   ffh_exchange_mask   sum[.][9] = closest_count where this shard holds the globally closest level;       then all-reduce SUM of sum,
 *                                                                                                          all-gather of fsum
 *   ffh_exchange_unpack reduced lanes + the gathered f64 sums added in rank order (= database order) -> summaries, in place */
int ffh_exchange_pack(ffh_ctx *ctx, const void *device_summaries, uint32_t n_guides, double *device_max, int32_t *device_sum, double *device_fsum);
int ffh_exchange_mask(ffh_ctx *ctx, const void *device_summaries, uint32_t n_guides, const double *device_max_reduced, int32_t *device_sum);
int ffh_exchange_unpack(ffh_ctx *ctx, void *device_summaries, uint32_t n_guides, const double *device_max_reduced, const int32_t *device_sum_reduced,
                        const double *device_fsum_all /* [world][n_guides][3] */, uint32_t world);

/* The exchange without a host round trip and without a second pass over the hits.  The ordered cut-off couples the shards only
 * for a guide whose positions reach maximumOffTargets somewhere along the shard order (crispr/CRISPRSiteOT.scala:39-46): for every
 * other guide the aggregates a shard computes on its own are already final.  So every shard first aggregates as if it were the
 * first one,
 *   ffh_finalize_shard        aggregates with prior 0 -> device_summaries [n_guides], shard totals (saturated at max_offtargets)
 *                             -> device_totals [n_guides];                                          then all-gather of the totals
 *   ffh_exchange_prior        device_prior[g] = min(sum of all_totals[r][g] over r < rank, clamp)
 *   ffh_finalize_shard_fixup  re-aggregates, with the prior, exactly the guides it changes anything for (prior > 0 and
 *                             prior + shard total >= max_offtargets) and overwrites their entries of device_summaries
 * and the three reduction collectives above follow.  Same results as ffh_shard_totals_device + ffh_finalize(prior) +
 * ffh_summaries_to_device, bit for bit.
 * ffh_use_stream(on = 1): the context issues its work on the caller's HIP stream (a hipStream_t; NULL is the default stream) --
 * the stream the caller's collectives are ordered with (PyTorch: torch.cuda.current_stream().cuda_stream) -- and the
 * device-buffer entry points of this section then neither wait for the device before nor for the stream after their kernels:
 * the exchange is one stream-ordered sequence.  The stream stays the caller's.  on = 0: back to the context's own stream. */
int ffh_use_stream(ffh_ctx *ctx, void *hip_stream, int on);
int ffh_finalize_shard(ffh_ctx *ctx, int max_offtargets, unsigned flags /* FFH_FINALIZE_JOST */, void *device_summaries, uint32_t *device_totals);
int ffh_exchange_prior(ffh_ctx *ctx, const uint32_t *device_all_totals /* [world][n_guides] */, uint32_t n_guides, uint32_t rank, uint32_t clamp,
                       uint32_t *device_prior);
int ffh_finalize_shard_fixup(ffh_ctx *ctx, int max_offtargets, unsigned flags, const uint32_t *device_prior, const uint32_t *device_totals,
                             void *device_summaries);

/* ---------------------------------------------------------------------------------------------------------
 * The bin-sharded discover as ONE library call (BASELINE.json configs[3]): every shard scans all
 * guides and aggregates them as if it were the first shard, then ONE all-gather moves every shard's 88-byte per-guide aggregates
 * (which hold its saturated position totals) to every rank and each rank folds them in shard order itself: the running totals are
 * every shard's prior, so that the ordered cut-off of CRISPRSiteOT.addOT / full (crispr/CRISPRSiteOT.scala:39-46) continues across
 * shards in database order; integer lanes add, maxima and the closest hit reduce, the f64 sums are added in shard order.  Only a
 * guide whose cut-off falls inside a shard that has a non-zero prior makes that shard aggregate it again (a second all-gather;
 * never taken by a guide set without OVERFLOW guides).  A shard that could not be scanned takes part with a status record: every
 * rank returns the error, none is left inside a collective.  The collectives
 * are issued BY THE LIBRARY on the contexts' streams -- RCCL over xGMI (librccl is opened on first use) -- so a JVM host
 * (GPUTraverser) or the C++ CLI gets the reduce without any framework above the C ABI.  Replaces, for N GPUs, the one
 * traverser chosen in modules/OffTargetDiscovery.scala:119-135.
 *   ffh_comm_create_rank   one process per GPU: `ctx` holds shard `rank` of `world`; id128 = the 128 bytes ffh_comm_unique_id
 *                          produced on one rank, distributed by the caller (MPI, torch.distributed, a file) -> ncclCommInitRank
 *   ffh_comm_create_local  one process drives all shards: ctxs[i] holds shard i (database order).  Distinct devices ->
 *                          ncclCommInitAll + grouped collectives; several shards on one device (a rehearsal of an N-way run on
 *                          one GPU: RCCL refuses duplicate devices) -> device-to-device copies and local reduction kernels.
 *                          FFH_COMM=copy in the environment forces the latter.
 * The contexts stay the caller's (destroy the communicator first).  Errors: ffh_comm_last_error (NULL: of a failed create).
 * ------------------------------------------------------------------------------------------------------- */
struct ffh_guide_summary;
typedef struct ffh_comm ffh_comm;
int ffh_comm_unique_id(void *id128);
int ffh_comm_create_rank(ffh_ctx *ctx, int rank, int world, const void *id128, ffh_comm **out);
int ffh_comm_create_local(ffh_ctx *const *ctxs, int n_shards, ffh_comm **out);
void ffh_comm_destroy(ffh_comm *comm);
const char *ffh_comm_last_error(const ffh_comm *comm);
int ffh_comm_world(const ffh_comm *comm);          /* shards in total */
int ffh_comm_first_shard(const ffh_comm *comm);    /* number of this process's first shard */
int ffh_comm_local_shards(const ffh_comm *comm);   /* shards held by this process */
int ffh_comm_transport(const ffh_comm *comm);      /* 0 copies (shards share a device), 1 RCCL ncclCommInitAll, 2 RCCL ncclCommInitRank */
/* Which form the exchange of ffh_discover_sharded / ffh_comm_exchange takes (every rank of the communicator must choose the same):
 *   0  ONE all-gather of every shard's 88-byte per-guide records, folded by every rank (world x n_guides x 88 bytes received per rank; one latency)
 *   1  by guide slices: an all-to-all (rank j receives everybody's records of slice j and folds them), the priors back in a second
 *      all-to-all, the folded slices all-gathered: three collectives, ~1 / world of the payload.  Bit-identical results.
 * Default 0; FFH_EXCHANGE=slice in the environment makes 1 the default of communicators created afterwards.  Which one is faster on
 * xGMI at which world size is a measurement nobody has made yet (no multi-GPU run exists): bench.py --gpus N reports both per-rank
 * exchange times when asked (FFH_EXCHANGE). */
int ffh_comm_set_exchange(ffh_comm *comm, int mode);
int ffh_comm_get_exchange(const ffh_comm *comm);
/* Page-locked host memory for buffers the CALLER hands to the library (summaries_out of ffh_discover_sharded / ffh_comm_exchange): the
 * copy-out of 100 000 summaries (8.8 MB) takes 0.17 ms into such a buffer and about twice that into pageable memory, where the runtime
 * stages it.  (The results of ffh_finalize / ffh_discover live in page-locked blocks of the context's own pool already.)  A JNI binding
 * wraps the pointer in a direct ByteBuffer (NewDirectByteBuffer).  NULL when the memory cannot be had; ffh_host_free(NULL) is a no-op. */
void *ffh_host_alloc(size_t bytes);
void ffh_host_free(void *p);

/* ffh_scan_bounded on every local shard (one host thread each) + the exchange; the reduced per-guide aggregates of ALL shards
 * land in summaries_out[n_guides] (host memory, pageable or -- faster -- from ffh_host_alloc; may be NULL on ranks that do not want them).  `guides`: host or device memory
 * (device: of every local shard's GPU, i.e. one local shard).  flags: FFH_FINALIZE_JOST. */
int ffh_discover_sharded(ffh_comm *comm, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets, unsigned flags,
                         struct ffh_guide_summary *summaries_out);
/* the exchange alone, after the caller scanned every local shard itself with the same guide set */
int ffh_comm_exchange(ffh_comm *comm, uint32_t n_guides, int max_offtargets, unsigned flags, struct ffh_guide_summary *summaries_out);
/* after ffh_discover_sharded: the retained hit list of a local shard under the cut-off continued from the shards before it
 * (ffh_finalize with the prior the exchange left on the device); concatenated in shard order the lists are the reference's.
 * flags: FFH_FINALIZE_NO_POSITIONS / _NO_HIT_SCORES / _JOST */
int ffh_comm_shard_lists(ffh_comm *comm, int local_shard, unsigned flags, ffh_result **out);
/* the reduced aggregates as they sit on a local shard's device (n_guides x ffh_guide_summary), valid until the next exchange */
int ffh_comm_device_summaries(ffh_comm *comm, int local_shard, const void **device_summaries);
/* host wall time of the last ffh_discover_sharded: the scans (slowest local shard), the exchange incl. the copy-out */
int ffh_comm_timings(const ffh_comm *comm, double *scan_ms, double *exchange_ms);

/* ---------------------------------------------------------------------------------------------------------
 * Several discover calls in flight against ONE resident database (round 6).
 * The reference's traverser serves one guide set at a time (reference/traverser/LinearTraverser.scala:59-130); a host that cuts a guide file
 * into batches pays every batch's host round trips with the device idle in between.  ffh_ctx_share_db makes a second context ON THE SAME
 * DEVICE that scans `owner`'s database through aliases of its device memory (targets, positions, the two scan images: nothing is copied);
 * everything a scan writes is the new context's own, on its own stream, so the two contexts may be driven from two host threads at once and
 * every call returns what it would return alone.  While a sharing context exists the owner refuses to load or rebuild its database
 * (FFH_E_STATE); destroy the sharing contexts before the owner.
 * ffh_pipe_*: the convenience on top -- `lanes` contexts (the owner + lanes - 1 sharing ones), one host thread each, a FIFO of submitted
 * guide batches.  ffh_pipe_submit copies the guides and returns a ticket at once; ffh_pipe_wait(ticket) blocks until that batch is done
 * and hands over its result (ffh_discover's, flags as there; free it with ffh_result_free).  The owner context must not be used directly
 * while the pipe exists.  ffh_pipe_destroy runs what is still queued, frees uncollected results and the sharing contexts.
 * ------------------------------------------------------------------------------------------------------- */
int ffh_ctx_share_db(ffh_ctx *owner, ffh_ctx **out);
typedef struct ffh_pipe ffh_pipe;
int ffh_pipe_create(ffh_ctx *owner, int lanes /* 1 .. 8 */, ffh_pipe **out);
int ffh_pipe_submit(ffh_pipe *pipe, const uint64_t *guides, uint32_t n_guides, int max_mismatch, int max_offtargets, unsigned flags, uint64_t *ticket);
int ffh_pipe_wait(ffh_pipe *pipe, uint64_t ticket, ffh_result **out);
const char *ffh_pipe_last_error(const ffh_pipe *pipe);
int ffh_pipe_lanes(const ffh_pipe *pipe);
void ffh_pipe_destroy(ffh_pipe *pipe);

/* The `score` path (modules/ScoreResults.scala:90-154): hit lists that already exist (re-read from a discover table)
 * are scored on the device with the same epilogue.  guide_offsets has n_guides+1 entries into hit_targets; the
 * lists are taken as they are (no cut-off, overflow = 0).  No database needs to be loaded.  The result carries
 * summaries, mismatches and per-hit CFD; its position arrays are empty. */
int ffh_score_lists(ffh_ctx *ctx, const uint64_t *guides, uint32_t n_guides, const uint64_t *guide_offsets,
                    const uint64_t *hit_targets, ffh_result **out);

/* ---------------------------------------------------------------------------------------------------------
 * Results: guides in input order; per guide the retained hit list in database order, already cut off.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct ffh_guide_summary {
    uint32_t n_hits;         /* retained off-target sequences (CRISPRSiteOT.offTargets.size) */
    uint32_t ot_count;       /* sum of their positions = the table's otCount column */
    uint32_t overflow;       /* CRISPRSiteOT.full after the scan -> OVERFLOW / OK */
    uint32_t hist[5];        /* ClosestHit: count-weighted histogram over 0..4 mismatches */
    uint32_t closest;        /* ClosestHit: smallest non-zero mismatch count, 0xFFFFFFFF = none ("UNK") */
    uint32_t closest_count;
    uint32_t in_genome;      /* DangerousSequences: occurrences at 0 mismatches */
    uint32_t n_scored;       /* hits that entered the CFD / Hsu2013 sums (mismatches != 0) */
    double cfd_max;          /* max over hits of pam*cfd, 0.0 if none (before the 0.023 print threshold) */
    double cfd_sum;          /* sum of pam*cfd*count in database order; specificity = 1/(1+cfd_sum) */
    double hsu_sum;          /* sum of Hsu2013 hit scores in database order; score = 100/(100+hsu_sum)*100 */
    double jost_max;         /* JostAndSantosCRISPRi: max hit activity, 0.0 if none (scoring/JostAndSantosCRISPRi.scala:43) */
    double jost_sum;         /* sum of activity*count in database order; specificity = 1/(1+jost_sum) (:42).  Both are 0
                                unless FFH_FINALIZE_JOST was passed (ffh_score_lists always fills them) */
} ffh_guide_summary;

uint32_t ffh_result_n_guides(const ffh_result *r);
uint64_t ffh_result_n_hits(const ffh_result *r);
uint64_t ffh_result_n_positions(const ffh_result *r);
int      ffh_result_scores_valid(const ffh_result *r);     /* CFD / Hsu2013 defined for this enzyme (Cas9 23-mer) */
const ffh_guide_summary *ffh_result_summaries(const ffh_result *r);      /* [n_guides] */
const uint64_t *ffh_result_guide_offsets(const ffh_result *r);           /* [n_guides+1] into the hit arrays */
const uint64_t *ffh_result_hit_targets(const ffh_result *r);             /* [n_hits] target longs incl. count */
const uint8_t  *ffh_result_hit_mismatches(const ffh_result *r);          /* [n_hits] */
const double   *ffh_result_hit_cfd(const ffh_result *r);                 /* [n_hits] pam*cfd, NaN where not scored */
const uint64_t *ffh_result_pos_offsets(const ffh_result *r);             /* [n_hits+1] into positions.  Not copied from the device:
                                                                            hit h owns (hit_targets[h] >> 48) positions, the offsets are
                                                                            folded from that on the host the first time they are asked for */
const uint64_t *ffh_result_positions(const ffh_result *r);               /* [n_positions] BitPosition longs */
void ffh_result_free(ffh_result *r);
/* Result arrays live in page-locked host memory (pooled per context).  FFH_PINNED_LIMIT_MB in the environment caps what ONE result
 * block may take; a discover whose lists need more returns FFH_E_NOMEM ("out of (pinned) host memory") and leaves the context usable. */

/* ---------------------------------------------------------------------------------------------------------
 * Instrumentation of the last ffh_scan/ffh_finalize on this context (HIP events on the context's stream).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct ffh_timings {
    double prepare_ms;        /* guide encode + candidate lists (CSR) of both images */
    double compare_ms;        /* the compare kernel (the dominant kernel): ONE launch per guide batch covering both images */
    double sort_ms;           /* hit ordering */
    double finalize_ms;       /* cut-off, scoring, aggregation, gathers */
    double total_scan_ms;     /* ffh_scan, first launch to last event */
    uint64_t n_raw_hits;      /* hits before the cut-off */
    uint64_t pairs_prefix;    /* full-length comparisons executed by the prefix pass */
    uint64_t pairs_suffix;    /* ... by the suffix pass */
    uint64_t items_prefix;    /* (bucket, guide) candidate entries enumerated */
    uint64_t items_suffix;
    uint64_t tiles_prefix;    /* work entries the compare launch walked for the prefix image (runs of buckets; field name kept from round 1) */
    uint64_t tiles_suffix;
    uint32_t compare_launches; /* guide batches (x slabs of a bounded scan) */
    int prefix_bases, prefix_radius, suffix_radius;
    uint32_t bounded_slabs;   /* 0: every guide met the whole shard; else the slabs of the bounded scan (ffh_scan_bounded) */
    uint32_t retired_guides;  /* guides that reached the limit before the last slab and were not scanned against it */
} ffh_timings;
int ffh_get_timings(const ffh_ctx *ctx, ffh_timings *out);

#ifdef __cplusplus
}
#endif
#endif
