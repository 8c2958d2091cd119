/*
 * flashfry_jni.c -- the JNI side of reference.traverser.GPUTraverser (jni/GPUTraverser.scala): one thin function per native method,
 * no logic of its own.  Every function forwards to the C ABI of include/flashfry_hip.h; the call sequence a scan makes
 *     create -> dbOpen -> discover -> resultOffsets / resultTargets / resultPosOffsets / resultPositions -> resultFree -> destroy
 * and, with several devices (bins sharded statically over the GPUs of the node, north star; property flashfry.gpu.devices),
 *     create x N -> dbOpenHeader + dbBins + dbBinBytes (bin cuts balanced by payload, BinaryHeader.scala:54) -> dbOpen x N (own bin range)
 *     -> createLocalComm -> discoverSharded -> shardLists x N (shard order = database order) -> result* -> resultFree x N -> commDestroy -> destroy x N
 * are exercised without a JVM by tests/test_jni_sequence.c (same order, same arguments, checked against the oracle).
 *
 * Build (needs a JDK for jni.h; this repository's image has none, so the file is compiled only where JAVA_HOME is set):
 *     make -C jni            -> jni/libflashfry_jni.so
 * The Scala object is `object GPUTraverser` in package reference.traverser: its methods are mangled with the module class suffix
 * "$" = _00024.
 */
#include <jni.h>
#include <stdint.h>

#include "../include/flashfry_hip.h"

#define FN(name) Java_reference_traverser_GPUTraverser_00024_##name
#define CTX(h) ((ffh_ctx *)(intptr_t)(h))
#define RES(h) ((const ffh_result *)(intptr_t)(h))
#define COMM(h) ((ffh_comm *)(intptr_t)(h))

/* ffh_create: Traverser.scan has no GPU argument; device 0 unless the property flashfry.gpu.device says otherwise (read on the Scala side) */
JNIEXPORT jlong JNICALL FN(create)(JNIEnv *e, jobject self, jint device, jint enzyme_index) {
    (void)e; (void)self;
    return (jlong)(intptr_t)ffh_create((int)device, (int)enzyme_index);
}

JNIEXPORT void JNICALL FN(destroy)(JNIEnv *e, jobject self, jlong ctx) {
    (void)e; (void)self;
    ffh_destroy(CTX(ctx));
}

/* ffh_db_open: header + BGZF body; bins [bin_begin, bin_end), 0 = to the last bin */
JNIEXPORT jint JNICALL FN(dbOpen)(JNIEnv *e, jobject self, jlong ctx, jstring path, jint bin_begin, jint bin_end) {
    (void)self;
    const char *p = (*e)->GetStringUTFChars(e, path, 0);
    if (!p) return FFH_E_NOMEM;
    const int rc = ffh_db_open(CTX(ctx), p, (uint32_t)bin_begin, (uint32_t)bin_end);
    (*e)->ReleaseStringUTFChars(e, path, p);
    return (jint)rc;
}

/* ffh_discover: the guides' longs are GuideIndex.guide unchanged (bitcoding/BitEncoding.scala:46-67); returns the ffh_result handle or 0 */
JNIEXPORT jlong JNICALL FN(discover)(JNIEnv *e, jobject self, jlong ctx, jlongArray guides, jint max_mismatch, jint max_offtargets) {
    (void)self;
    const jsize n = (*e)->GetArrayLength(e, guides);
    jlong *g = (*e)->GetLongArrayElements(e, guides, 0);
    if (!g) return 0;
    ffh_result *r = 0;
    const int rc = ffh_discover(CTX(ctx), (const uint64_t *)g, (uint32_t)n, (int)max_mismatch, (int)max_offtargets, FFH_FINALIZE_NO_HIT_SCORES /* CRISPRHit carries the sequence and its positions; scores are ScoreModel work */, &r);
    (*e)->ReleaseLongArrayElements(e, guides, g, JNI_ABORT);
    return rc ? 0 : (jlong)(intptr_t)r;
}

/* ffh_pipe_*: the batches of a large guide set in flight against the one resident database (round 6) */
#define PIPE(h) ((ffh_pipe *)(intptr_t)(h))
JNIEXPORT jlong JNICALL FN(pipeCreate)(JNIEnv *e, jobject self, jlong ctx, jint lanes) {
    (void)e; (void)self;
    ffh_pipe *p = 0;
    return ffh_pipe_create(CTX(ctx), (int)lanes, &p) ? 0 : (jlong)(intptr_t)p;
}
JNIEXPORT jlong JNICALL FN(pipeSubmit)(JNIEnv *e, jobject self, jlong pipe, jlongArray guides, jint max_mismatch, jint max_offtargets) {   /* the ticket, 0 on error */
    (void)self;
    const jsize n = (*e)->GetArrayLength(e, guides);
    jlong *g = (*e)->GetLongArrayElements(e, guides, 0);
    if (!g) return 0;
    uint64_t ticket = 0;
    const int rc = ffh_pipe_submit(PIPE(pipe), (const uint64_t *)g, (uint32_t)n, (int)max_mismatch, (int)max_offtargets, FFH_FINALIZE_NO_HIT_SCORES, &ticket);   /* (the guides are copied) */
    (*e)->ReleaseLongArrayElements(e, guides, g, JNI_ABORT);
    return rc ? 0 : (jlong)ticket;
}
JNIEXPORT jlong JNICALL FN(pipeWait)(JNIEnv *e, jobject self, jlong pipe, jlong ticket) {   /* the ffh_result handle or 0 */
    (void)e; (void)self;
    ffh_result *r = 0;
    return ffh_pipe_wait(PIPE(pipe), (uint64_t)ticket, &r) ? 0 : (jlong)(intptr_t)r;
}
JNIEXPORT jstring JNICALL FN(pipeLastError)(JNIEnv *e, jobject self, jlong pipe) {
    (void)self;
    return (*e)->NewStringUTF(e, ffh_pipe_last_error(PIPE(pipe)));
}
JNIEXPORT void JNICALL FN(pipeDestroy)(JNIEnv *e, jobject self, jlong pipe) {
    (void)e; (void)self;
    ffh_pipe_destroy(PIPE(pipe));
}

static jlongArray to_jlongs(JNIEnv *e, const uint64_t *p, uint64_t n) {
    if (n > 0x7FFFFFF0ull) return 0;   /* a Java array holds < 2^31 elements: the caller splits the guide set before that */
    jlongArray a = (*e)->NewLongArray(e, (jsize)n);
    if (a && n) (*e)->SetLongArrayRegion(e, a, 0, (jsize)n, (const jlong *)p);
    return a;
}

JNIEXPORT jlongArray JNICALL FN(resultOffsets)(JNIEnv *e, jobject self, jlong res) {      /* [n_guides + 1] into the hit arrays */
    (void)self;
    return to_jlongs(e, ffh_result_guide_offsets(RES(res)), (uint64_t)ffh_result_n_guides(RES(res)) + 1);
}
JNIEXPORT jlongArray JNICALL FN(resultTargets)(JNIEnv *e, jobject self, jlong res) {      /* [n_hits] target longs incl. count */
    (void)self;
    return to_jlongs(e, ffh_result_hit_targets(RES(res)), ffh_result_n_hits(RES(res)));
}
JNIEXPORT jlongArray JNICALL FN(resultPosOffsets)(JNIEnv *e, jobject self, jlong res) {   /* [n_hits + 1] into positions */
    (void)self;
    return to_jlongs(e, ffh_result_pos_offsets(RES(res)), ffh_result_n_hits(RES(res)) + 1);
}
JNIEXPORT jlongArray JNICALL FN(resultPositions)(JNIEnv *e, jobject self, jlong res) {    /* [n_positions] BitPosition longs */
    (void)self;
    return to_jlongs(e, ffh_result_positions(RES(res)), ffh_result_n_positions(RES(res)));
}

JNIEXPORT void JNICALL FN(resultFree)(JNIEnv *e, jobject self, jlong res) {
    (void)e; (void)self;
    ffh_result_free((ffh_result *)(intptr_t)res);
}

/* ffh_last_error: ctx == 0 asks for the message of a failed create (kept per calling thread) */
JNIEXPORT jstring JNICALL FN(lastError)(JNIEnv *e, jobject self, jlong ctx) {
    (void)self;
    const char *m = ffh_last_error(CTX(ctx));
    return (*e)->NewStringUTF(e, m ? m : "");
}

/* ---- several GPUs: one context per device, bins sharded statically, the exchange inside the library (ffh_comm_*) ---- */

/* ffh_db_open_header: the header alone -- enzyme, contigs, bins and their payload bytes (what the bin cuts are balanced by) */
JNIEXPORT jint JNICALL FN(dbOpenHeader)(JNIEnv *e, jobject self, jlong ctx, jstring path) {
    (void)self;
    const char *p = (*e)->GetStringUTFChars(e, path, 0);
    if (!p) return FFH_E_NOMEM;
    const int rc = ffh_db_open_header(CTX(ctx), p);
    (*e)->ReleaseStringUTFChars(e, path, p);
    return (jint)rc;
}
JNIEXPORT jint JNICALL FN(dbBins)(JNIEnv *e, jobject self, jlong ctx) {
    (void)e; (void)self;
    ffh_db_info info;
    return ffh_db_info_get(CTX(ctx), &info) ? 0 : (jint)info.n_bins;
}
JNIEXPORT jlong JNICALL FN(dbBinBytes)(JNIEnv *e, jobject self, jlong ctx, jint bin) {
    (void)e; (void)self;
    return (jlong)ffh_db_bin_bytes(CTX(ctx), (uint32_t)bin);
}

/* ffh_comm_create_local: ctxs[i] holds shard i (database order); distinct devices -> RCCL (ncclCommInitAll), a device named twice ->
 * device copies.  Returns the ffh_comm handle or 0 (lastError(0) has the message). */
JNIEXPORT jlong JNICALL FN(createLocalComm)(JNIEnv *e, jobject self, jlongArray ctxs) {
    (void)self;
    const jsize n = (*e)->GetArrayLength(e, ctxs);
    if (n < 1 || n > 64) return 0;
    jlong *h = (*e)->GetLongArrayElements(e, ctxs, 0);
    if (!h) return 0;
    ffh_ctx *c[64];
    for (jsize i = 0; i < n; ++i) c[i] = CTX(h[i]);
    (*e)->ReleaseLongArrayElements(e, ctxs, h, JNI_ABORT);
    ffh_comm *cm = 0;
    return ffh_comm_create_local(c, (int)n, &cm) ? 0 : (jlong)(intptr_t)cm;
}
JNIEXPORT void JNICALL FN(commDestroy)(JNIEnv *e, jobject self, jlong comm) {
    (void)e; (void)self;
    ffh_comm_destroy(COMM(comm));
}
/* ffh_discover_sharded: every shard scans all guides, the shards exchange their aggregates (ordered cut-off continued in shard order);
 * the reduced per-guide aggregates are not taken to the JVM here (ScoreModel work happens on the hit lists); 0 = ok */
JNIEXPORT jint JNICALL FN(discoverSharded)(JNIEnv *e, jobject self, jlong comm, jlongArray guides, jint max_mismatch, jint max_offtargets) {
    (void)self;
    const jsize n = (*e)->GetArrayLength(e, guides);
    jlong *g = (*e)->GetLongArrayElements(e, guides, 0);
    if (!g) return FFH_E_NOMEM;
    const int rc = ffh_discover_sharded(COMM(comm), (const uint64_t *)g, (uint32_t)n, (int)max_mismatch, (int)max_offtargets, 0u, 0);
    (*e)->ReleaseLongArrayElements(e, guides, g, JNI_ABORT);
    return (jint)rc;
}
/* ffh_comm_shard_lists: the retained hits of local shard `shard` under the cut-off continued from the shards before it; an ffh_result
 * handle for the result* accessors, or 0 */
JNIEXPORT jlong JNICALL FN(shardLists)(JNIEnv *e, jobject self, jlong comm, jint shard) {
    (void)e; (void)self;
    ffh_result *r = 0;
    return ffh_comm_shard_lists(COMM(comm), (int)shard, FFH_FINALIZE_NO_HIT_SCORES, &r) ? 0 : (jlong)(intptr_t)r;
}
JNIEXPORT jstring JNICALL FN(commLastError)(JNIEnv *e, jobject self, jlong comm) {
    (void)self;
    const char *m = ffh_comm_last_error(COMM(comm));
    return (*e)->NewStringUTF(e, m ? m : "");
}
