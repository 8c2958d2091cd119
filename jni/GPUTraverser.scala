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
  // several GPUs (flashfry.gpu.devices): one context per device, the shards' exchange inside the library
  @native private def dbOpenHeader(ctx: Long, path: String): Int                        // ffh_db_open_header
  @native private def dbBins(ctx: Long): Int                                            // ffh_db_info_get: n_bins
  @native private def dbBinBytes(ctx: Long, bin: Int): Long                             // ffh_db_bin_bytes
  @native private def createLocalComm(ctxs: Array[Long]): Long                          // ffh_comm_create_local: ffh_comm*, 0 on error
  @native private def commDestroy(comm: Long): Unit
  @native private def discoverSharded(comm: Long, guides: Array[Long], maxMismatch: Int, maxOffTargets: Int): Int // ffh_discover_sharded
  @native private def shardLists(comm: Long, shard: Int): Long                          // ffh_comm_shard_lists: ffh_result*, 0 on error
  @native private def commLastError(comm: Long): String
  // the batches of a large guide set in flight against the one resident database (ffh_pipe_*: sharing contexts, one host thread each)
  @native private def pipeCreate(ctx: Long, lanes: Int): Long                           // ffh_pipe*, 0 on error
  @native private def pipeSubmit(pipe: Long, guides: Array[Long], maxMismatch: Int, maxOffTargets: Int): Long // ticket, 0 on error
  @native private def pipeWait(pipe: Long, ticket: Long): Long                          // ffh_result*, 0 on error
  @native private def pipeLastError(pipe: Long): String
  @native private def pipeDestroy(pipe: Long): Unit

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

  /** the devices of a run: -Dflashfry.gpu.devices=0,1,2,3,4,5,6,7 (a device may be named twice: two bin shards on it), else the one of
    * -Dflashfry.gpu.device (default 0) */
  def devices: Array[Int] = Option(System.getProperty("flashfry.gpu.devices")) match {
    case Some(s) if s.trim.nonEmpty =>
      val ids = s.split(",").map(_.trim)
      ids.foreach { t =>
        if (t.isEmpty || !t.forall(_.isDigit)) throw new IllegalArgumentException("flashfry.gpu.devices: '" + s + "' is not a comma-separated list of device numbers")
      }
      ids.map(_.toInt)
    case _ => Array(Integer.getInteger("flashfry.gpu.device", 0).intValue)
  }

  /** contiguous bin ranges, balanced by the bins' payload bytes (what BinaryHeader.uncompressedSize records per bin,
    * reference/binary/BinaryHeader.scala:54): cut(r) = the first bin of shard r, cut(n) = the number of bins */
  def binCuts(binBytes: Array[Long], n: Int): Array[Int] = {
    val total = binBytes.map(_.toDouble).sum
    val cut = Array.fill(n + 1)(binBytes.length)
    cut(0) = 0
    var run = 0.0
    var r = 1
    var b = 0
    while (b < binBytes.length && r < n) {
      run += binBytes(b).toDouble
      while (r < n && run >= total * r.toDouble / n.toDouble) { cut(r) = b + 1; r += 1 }
      b += 1
    }
    cut
  }

  /** replays the retained hits of one result (the whole database, or one bin shard) into the aggregator, guide by guide in database
    * order: exactly the updateOT sequence the CPU traversers produce for these bins.  The lists are already cut off by the library
    * (ordered cut-off, CRISPRSiteOT.scala:39-46, continued across shards), so updateOT's own `full` test never rejects a hit and
    * the overflow callback fires on the same hit it would have fired on */
  private def replay(res: Long, ctx: Long, batch: Array[crispr.GuideIndex], aggregator: ResultsAggregator) {
    val off = need(resultOffsets(res), "guide offsets", ctx)
    val tg = need(resultTargets(res), "hit targets", ctx)
    val po = need(resultPosOffsets(res), "position offsets", ctx)
    val ps = need(resultPositions(res), "positions", ctx)
    batch.indices.foreach { g =>
      var h = off(g).toInt
      val end = off(g + 1).toInt
      while (h < end) {
        aggregator.updateOT(batch(g), new CRISPRHit(tg(h), java.util.Arrays.copyOfRange(ps, po(h).toInt, po(h + 1).toInt)))
        h += 1
      }
    }
    Traverser.allTargetsAndPositions += ps.length
  }

  def scan(binaryFile: File,
           header: BinaryHeader,
           traversal: BinTraversal,
           aggregator: ResultsAggregator,
           maxMismatch: Int,
           configuration: ParameterPack,
           bitCoder: BitEncoding,
           posCoder: BitPosition) {

    var devs = devices
    val enzyme = ParameterPack.parameterPackToIndex(configuration)
    val path = binaryFile.getAbsolutePath // the body file; the library reads <path>.header itself (BinaryHeader.scala:115-160)
    var ctxs = new Array[Long](devs.length)
    var comm = 0L
    try {
      ctxs(0) = create(devs(0), enzyme)
      if (ctxs(0) == 0) throw new IllegalStateException("GPUTraverser: " + lastError(0))
      // one GPU: the whole database; several: shard i holds the bins [cut(i), cut(i + 1)) (static, contiguous, balanced by payload)
      var cut = Array(0, 0)
      if (devs.length > 1) {
        if (dbOpenHeader(ctxs(0), path) != 0) throw new IllegalStateException("GPUTraverser: " + lastError(ctxs(0)))
        val all = binCuts(Array.tabulate(dbBins(ctxs(0)))(b => dbBinBytes(ctxs(0), b)), devs.length)
        // more devices than bins with a payload (a small database, an empty one): the shards that would be empty are dropped -- an empty
        // bin range is not a database (ADVICE r4); the devices behind them stay idle
        val keep = devs.indices.filter(i => all(i + 1) > all(i))
        if (keep.length <= 1) devs = devs.take(1)
        else { cut = (keep.map(all(_)) :+ all.last).toArray; devs = keep.map(devs(_)).toArray }   // (dropped ranges are empty: the kept ones stay contiguous)
        ctxs = ctxs(0) +: new Array[Long](devs.length - 1)
      }
      (1 until devs.length).foreach { i =>
        ctxs(i) = create(devs(i), enzyme)
        if (ctxs(i) == 0) throw new IllegalStateException("GPUTraverser: " + lastError(0))
      }
      if (devs.length == 1) {
        if (dbOpen(ctxs(0), path, 0, 0) != 0) throw new IllegalStateException("GPUTraverser: " + lastError(ctxs(0)))
      } else {
        // the loads are independent: one host thread per device; every load runs to its end, then ONE exception names what failed
        val failed = devs.indices.par.map { i =>
          if (dbOpen(ctxs(i), path, cut(i), cut(i + 1)) != 0) Some("device " + devs(i) + " (bins " + cut(i) + " .. " + cut(i + 1) + "): " + lastError(ctxs(i))) else None
        }.seq.flatten
        if (failed.nonEmpty) throw new IllegalStateException("GPUTraverser: " + failed.mkString("; "))
        comm = createLocalComm(ctxs)
        if (comm == 0) throw new IllegalStateException("GPUTraverser: " + lastError(0))
      }

      val guides = aggregator.indexedGuides // GuideIndex(guide, index), sorted by start (ResultsAggregator.scala:34-48)
      if (guides.nonEmpty) {
        // maximumOffTargets: the same for every guide of a run (OffTargetDiscovery.scala:100-102); needs CRISPRSiteOT.overflowValue
        val maxOffTargets = aggregator.wrappedGuides.head.otSite.overflowValue

        val batches = guides.grouped(guidesPerCall(maxOffTargets)).toArray
        // One device, several batches: two calls in flight (-Dflashfry.gpu.lanes, default 2).  The reference's traverser serves one
        // aggregator at a time (LinearTraverser.scala:59-130); here batch k + 1 is scanned while batch k's lists cross the link and are
        // replayed into the aggregator.  Results are collected -- and replayed -- in submission order, so updateOT sees the sequence it
        // would see from sequential calls.
        val lanes = math.max(1, Integer.getInteger("flashfry.gpu.lanes", 2).intValue)
        if (devs.length == 1 && batches.length > 1 && lanes > 1) {
          val pipe = pipeCreate(ctxs(0), math.min(lanes, 8))
          if (pipe == 0) throw new IllegalStateException("GPUTraverser: " + lastError(ctxs(0)))
          try {
            val tickets = batches.map { batch =>
              val t = pipeSubmit(pipe, batch.map(_.guide), maxMismatch, maxOffTargets)
              if (t == 0) throw new IllegalStateException("GPUTraverser: a guide batch could not be submitted")
              t
            }
            batches.indices.foreach { k =>
              val res = pipeWait(pipe, tickets(k))
              if (res == 0) throw new IllegalStateException("GPUTraverser: " + pipeLastError(pipe))
              try replay(res, ctxs(0), batches(k), aggregator) finally resultFree(res)
            }
          } finally pipeDestroy(pipe)
        } else batches.foreach { batch =>
          if (devs.length == 1) {
            val res = discover(ctxs(0), batch.map(_.guide), maxMismatch, maxOffTargets)
            if (res == 0) throw new IllegalStateException("GPUTraverser: " + lastError(ctxs(0)))
            try replay(res, ctxs(0), batch, aggregator) finally resultFree(res)
          } else {
            // every shard scans all guides of the batch; the library continues the ordered cut-off across the shards and reduces the
            // aggregates (RCCL over xGMI between distinct devices).  The shards' lists, replayed in shard order, are the database-order
            // stream of the single traverser: a guide's hits of shard 0, then of shard 1, ...
            if (discoverSharded(comm, batch.map(_.guide), maxMismatch, maxOffTargets) != 0)
              throw new IllegalStateException("GPUTraverser: " + commLastError(comm))
            devs.indices.foreach { i =>
              val res = shardLists(comm, i)
              if (res == 0) throw new IllegalStateException("GPUTraverser: " + commLastError(comm))
              try replay(res, ctxs(i), batch, aggregator) finally resultFree(res)
            }
          }
        }
      }
    } finally {
      if (comm != 0) commDestroy(comm)
      ctxs.foreach(c => if (c != 0) destroy(c))
    }
  }
}
