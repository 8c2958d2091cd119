/*
 * GPUTraverser.scala -- the Traverser a FlashFry maintainer adds next to SeekTraverser / LinearTraverser
 * (src/main/scala/reference/traverser/), selected in OffTargetDiscovery.runWithOptions (modules/OffTargetDiscovery.scala:119-135)
 * by a new --gpu flag:
 *
 *     case _ if useGpu => {
 *       guideStorage.setTraversalOverFlowCallback(_ => ())      // updateOT calls the callback when a guide fills up (ResultsAggregator.scala:61-69)
 *       GPUTraverser.scan(new File(binaryOTFile), header, null, guideStorage, maxMismatch, header.inputParameterPack, header.bitCoder, header.bitPosition)
 *     }
 *
 * It implements trait Traverser (reference/traverser/Traverser.scala:38-61) with the same contract as the two CPU traversers:
 * every (guide, target) with BitEncoding.mismatches <= maxMismatch reaches aggregator.updateOT in database order, and a guide stops
 * receiving hits once CRISPRSiteOT.full (crispr/CRISPRSiteOT.scala:39-46).  The scan itself runs in libflashfry_hip.so through
 * jni/flashfry_jni.c; the BinTraversal argument is not used (candidate generation happens on the device).
 *
 * ONE change to the reference is needed: CRISPRSiteOT keeps maximumOffTargets as a constructor parameter (crispr/CRISPRSiteOT.scala:31,
 * `overflow: Int`), so it cannot be read back.  Add to that class:      val overflowValue: Int = overflow
 *
 * UNVERIFIED: never compiled or run (this repository's image has no JDK and no scalac).  What is tested is the call sequence
 * below replayed as plain C against a database file and the oracle (tests/test_jni_sequence.c); treat this file and
 * jni/flashfry_jni.c as the binding's specification until they have been built against a JDK.
 */
package reference.traverser

import java.io.File

import bitcoding.{BitEncoding, BitPosition}
import com.typesafe.scalalogging.LazyLogging
import crispr.{CRISPRHit, ResultsAggregator}
import reference.binary.BinaryHeader
import reference.traversal.BinTraversal
import standards.ParameterPack

object GPUTraverser extends Traverser with LazyLogging {
  System.loadLibrary("flashfry_jni") // libflashfry_jni.so, linked against libflashfry_hip.so

  @native private def create(device: Int, enzymeIndex: Int): Long
  @native private def destroy(ctx: Long): Unit
  @native private def dbOpen(ctx: Long, path: String, binBegin: Int, binEnd: Int): Int
  @native private def discover(ctx: Long, guides: Array[Long], maxMismatch: Int, maxOffTargets: Int): Long // ffh_result*, 0 on error
  @native private def resultOffsets(res: Long): Array[Long]    // ffh_result_guide_offsets: [guides + 1]
  @native private def resultTargets(res: Long): Array[Long]    // ffh_result_hit_targets
  @native private def resultPosOffsets(res: Long): Array[Long] // ffh_result_pos_offsets: [hits + 1]
  @native private def resultPositions(res: Long): Array[Long]  // ffh_result_positions
  @native private def resultFree(res: Long): Unit
  @native private def lastError(ctx: Long): String

  /** guides per native call, sized so that no returned Array[Long] can reach the 2^31 elements a JVM array holds: a guide keeps
    * fewer than maxOffTargets + 32767 positions (the hit that crosses the limit is kept whole, BlockReader.scala:147-153 caps a
    * target at 32767 positions) and never more hits than positions */
  def guidesPerCall(maxOffTargets: Int): Int =
    math.max(1L, math.min(200000L, (1L << 30) / (maxOffTargets.toLong + 32767L + 1L))).toInt

  /** a native accessor returns null when the array would not fit a JVM array or could not be allocated */
  private def need(a: Array[Long], what: String, ctx: Long): Array[Long] = {
    if (a == null) throw new IllegalStateException("GPUTraverser: " + what + " could not be handed to the JVM (" + lastError(ctx) + ")")
    a
  }

  def scan(binaryFile: File,
           header: BinaryHeader,
           traversal: BinTraversal,
           aggregator: ResultsAggregator,
           maxMismatch: Int,
           configuration: ParameterPack,
           bitCoder: BitEncoding,
           posCoder: BitPosition) {

    val device = Integer.getInteger("flashfry.gpu.device", 0).intValue
    val ctx = create(device, ParameterPack.parameterPackToIndex(configuration))
    if (ctx == 0) throw new IllegalStateException("GPUTraverser: " + lastError(0))
    try {
      // the database path is the body file; the library reads <path>.header itself (BinaryHeader.scala:115-160)
      if (dbOpen(ctx, binaryFile.getAbsolutePath, 0, 0) != 0) throw new IllegalStateException("GPUTraverser: " + lastError(ctx))

      val guides = aggregator.indexedGuides // GuideIndex(guide, index), sorted by start (ResultsAggregator.scala:34-48)
      if (guides.nonEmpty) {
        // maximumOffTargets: the same for every guide of a run (OffTargetDiscovery.scala:100-102); needs CRISPRSiteOT.overflowValue
        val maxOffTargets = aggregator.wrappedGuides.head.otSite.overflowValue

        guides.grouped(guidesPerCall(maxOffTargets)).foreach { batch =>
          val res = discover(ctx, batch.map(_.guide), maxMismatch, maxOffTargets)
          if (res == 0) throw new IllegalStateException("GPUTraverser: " + lastError(ctx))
          try {
            val off = need(resultOffsets(res), "guide offsets", ctx)
            val tg = need(resultTargets(res), "hit targets", ctx)
            val po = need(resultPosOffsets(res), "position offsets", ctx)
            val ps = need(resultPositions(res), "positions", ctx)
            // replay in database order: exactly the updateOT sequence the CPU traversers produce.  The lists are already cut off by
            // the library (ordered cut-off, CRISPRSiteOT.scala:39-46), so updateOT's own `full` test never rejects a hit and the
            // overflow callback fires on the same hit it would have fired on
            batch.indices.foreach { g =>
              var h = off(g).toInt
              val end = off(g + 1).toInt
              while (h < end) {
                aggregator.updateOT(batch(g), new CRISPRHit(tg(h), java.util.Arrays.copyOfRange(ps, po(h).toInt, po(h + 1).toInt)))
                h += 1
              }
            }
            Traverser.allTargetsAndPositions += ps.length
          } finally resultFree(res)
        }
      }
    } finally destroy(ctx)
  }
}
