"""Multi-GPU plumbing for the bin-sharded discover: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The database bins are sharded statically and contiguously, rank r owning bins [b_r, b_{r+1}); every rank scans all
guides against its shard.  Two small exchanges per discover:
  1. all-gather of the per-guide position totals of every shard (u32[G]) -> each rank's prior total, so that the
     ordered cut-off of CRISPRSiteOT.addOT/full (crispr/CRISPRSiteOT.scala:39-46) continues across shards in
     database order;
  2. reduction of the per-guide aggregates (integer lanes summed exactly, f64 sums combined in rank order,
     cfd_max / overflow by max, closest hit by min + masked count).
Hit lists stay on their rank; concatenated in rank order they are the reference's hit list.

Two forms of the same exchange:
  * native_comm(ctx) -> capi.Comm: the collectives are issued INSIDE libflashfry_hip (ffh_comm_*, csrc/ffh_comm.hpp: RCCL on the
    context's stream); torch.distributed only carries the 128-byte unique id once.  What bench.py --gpus N times, what the C++ CLI
    (--gpus N, ncclCommInitAll) and a JVM host call.
  * DeviceExchange: library kernels + torch.distributed collectives on torch's stream (any backend: the gloo tests on CPU tensors,
    several ranks sharing one GPU).
"""
import numpy as np


def native_comm(ctx, group=None):
    """one rank per GPU: an ffh_comm over RCCL for `ctx` (this rank's shard); rank 0 makes the unique id, torch.distributed hands it round"""
    import torch.distributed as dist
    from . import capi
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    uid = [capi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return capi.Comm.rank(ctx, rank, world, uid[0])


def shard_bins(uncompressed_bytes, world):
    """contiguous bin ranges balanced by payload bytes (BinaryHeader's uncompressedSize), [(begin, end)] * world"""
    sizes = np.asarray(uncompressed_bytes, dtype=np.float64)
    csum = np.concatenate([[0.0], np.cumsum(sizes)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(csum, total * r / world, side="left")))
    cuts.append(len(sizes))
    cuts = [min(max(c, 0), len(sizes)) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def prior_totals(local_totals, clamp, device=None, group=None):
    """exclusive prefix over ranks of the per-guide shard totals, saturated at `clamp` (the totals themselves must
    already be saturated at clamp: once a lower shard reaches the limit nothing later is retained)"""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    t = torch.as_tensor(np.asarray(local_totals).astype(np.int64), device=device)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)
    prior = torch.zeros_like(t)
    for r in range(rank):
        prior += gathered[r]
    return torch.clamp(prior, max=int(clamp)).cpu().numpy().astype(np.uint32)


def allreduce_summaries(summ, device=None, group=None):
    """in-place reduction of a structured per-guide summary array (flashfry_amd.capi.SUMMARY_DTYPE) over all ranks"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = len(summ)
    ints = np.zeros((n, 9), dtype=np.int64)
    ints[:, 0] = summ["n_hits"]; ints[:, 1] = summ["ot_count"]; ints[:, 2:7] = summ["hist"]
    ints[:, 7] = summ["in_genome"]; ints[:, 8] = summ["n_scored"]
    ti = torch.as_tensor(ints, device=device)
    dist.all_reduce(ti, op=dist.ReduceOp.SUM, group=group)
    mx = torch.as_tensor(np.stack([summ["overflow"].astype(np.float64), summ["cfd_max"], summ["jost_max"]], 1), device=device)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    closest = torch.as_tensor(summ["closest"].astype(np.int64), device=device)
    gmin = closest.clone()
    dist.all_reduce(gmin, op=dist.ReduceOp.MIN, group=group)
    cc = torch.as_tensor(summ["closest_count"].astype(np.int64), device=device)
    cc = torch.where(closest == gmin, cc, torch.zeros_like(cc))
    dist.all_reduce(cc, op=dist.ReduceOp.SUM, group=group)
    # f64 sums: gather and add in rank order (= database order of the shards) so the result does not depend on the
    # collective's internal reduction order
    fl = torch.as_tensor(np.stack([summ["cfd_sum"], summ["hsu_sum"], summ["jost_sum"]], 1), device=device)
    parts = [torch.empty_like(fl) for _ in range(world)]
    dist.all_gather(parts, fl, group=group)
    acc = parts[0].clone()
    for r in range(1, world):
        acc += parts[r]
    ti, mx, gmin, cc, acc = ti.cpu().numpy(), mx.cpu().numpy(), gmin.cpu().numpy(), cc.cpu().numpy(), acc.cpu().numpy()
    summ["n_hits"] = ti[:, 0]; summ["ot_count"] = ti[:, 1]; summ["hist"] = ti[:, 2:7]
    summ["in_genome"] = ti[:, 7]; summ["n_scored"] = ti[:, 8]
    summ["overflow"] = mx[:, 0].astype(np.uint32); summ["cfd_max"] = mx[:, 1]; summ["jost_max"] = mx[:, 2]
    summ["closest"] = gmin.astype(np.uint32); summ["closest_count"] = cc
    summ["cfd_sum"] = acc[:, 0]; summ["hsu_sum"] = acc[:, 1]; summ["jost_sum"] = acc[:, 2]
    return summ


class DeviceExchange:
    """The two exchanges of a sharded discover on device memory (no host round trip): buffers are allocated once.

    step(ctx, max_offtargets) runs after ctx.scan(...) on every rank: shard totals -> all-gather -> prior totals ->
    ffh_finalize (aggregates only) -> reduction of the aggregates; returns the reduced summaries as a
    uint8 [G, itemsize] device tensor (view it with summaries_numpy())."""

    def __init__(self, n_guides, device, group=None, itemsize=88):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.device = torch, dist, group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.G, self.itemsize = n_guides, itemsize
        assert itemsize == 88, "layout of ffh_guide_summary: 12 x u32 then 5 x f64"
        self.totals = torch.zeros(n_guides, dtype=torch.int32, device=device)
        self.all_totals = torch.zeros(self.world * n_guides, dtype=torch.int32, device=device)
        self.prior = torch.zeros(n_guides, dtype=torch.int32, device=device)
        self.summ = torch.zeros(n_guides * itemsize, dtype=torch.uint8, device=device)
        self.fsum_all = torch.zeros(self.world * n_guides * 3, dtype=torch.float64, device=device)
        self.mx = torch.zeros(n_guides * 4, dtype=torch.float64, device=device)
        self.sums = torch.zeros(n_guides * 10, dtype=torch.int32, device=device)
        self.fsum = torch.zeros(n_guides * 3, dtype=torch.float64, device=device)

    def _all_gather(self, out_flat, inp):
        # chunk views of one flat buffer: works with every backend (RCCL gathers in place, gloo copies)
        self.dist.all_gather(list(out_flat.chunk(self.world)), inp, group=self.group)

    def prior_totals(self, ctx, max_offtargets):
        ctx.shard_totals_device(self.totals.data_ptr(), max_offtargets)
        return self.prior_from_totals(max_offtargets)

    def prior_from_totals(self, max_offtargets):
        """self.totals (this shard's saturated per-guide totals) -> self.prior (saturated sum over the lower-ranked shards)"""
        torch = self.torch
        self._all_gather(self.all_totals, self.totals)
        if self.rank:
            lower = self.all_totals.view(self.world, self.G)[: self.rank].to(torch.int64).sum(0)
            self.prior.copy_(torch.clamp(lower, max=int(max_offtargets)).to(torch.int32))
        else:
            self.prior.zero_()
        return self.prior

    def reduce_summaries(self):
        """in place on self.summ, three collectives: (1) MAX over [overflow, cfd_max, jost_max, -closest] (the min of the closest
        level rides along negated; every value is exact in f64), (2) SUM over the integer lanes plus the closest-hit count masked
        to the winning level, (3) all-gather of the three f64 sums, added in rank order (= database order of the shards:
        deterministic, like the host path)"""
        torch, dist, G = self.torch, self.dist, self.G
        i32 = self.summ.view(torch.int32).view(G, 22)
        f64 = self.summ.view(torch.float64).view(G, 11)
        closest = (i32[:, 8].to(torch.int64) & 0xFFFFFFFF).to(torch.float64)          # 0xFFFFFFFF = none
        mx = torch.stack([i32[:, 2].to(torch.float64), f64[:, 6], f64[:, 9], -closest], 1).contiguous()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        gmin = -mx[:, 3]
        cc = torch.where(closest == gmin, i32[:, 9], torch.zeros_like(i32[:, 9]))
        sums = torch.cat([i32[:, [0, 1, 3, 4, 5, 6, 7, 10, 11]], cc.unsqueeze(1)], 1).contiguous()  # n_hits, ot_count, hist[5], in_genome, n_scored, closest_count
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
        fl = f64[:, [7, 8, 10]].contiguous().view(-1)                                  # cfd_sum, hsu_sum, jost_sum
        self._all_gather(self.fsum_all, fl)
        parts = self.fsum_all.view(self.world, G, 3)
        acc = parts[0].clone()
        for r in range(1, self.world):
            acc += parts[r]
        i32[:, [0, 1, 3, 4, 5, 6, 7, 10, 11]] = sums[:, :9]
        i32[:, 9] = sums[:, 9]
        i32[:, 2] = mx[:, 0].to(torch.int32)
        f64[:, 6] = mx[:, 1]
        f64[:, 9] = mx[:, 2]
        gm = gmin.to(torch.int64)
        i32[:, 8] = torch.where(gm > 0x7FFFFFFF, gm - (1 << 32), gm).to(torch.int32)
        f64[:, [7, 8, 10]] = acc
        return self.summ

    def reduce_summaries_sliced(self):
        """reduce_summaries by GUIDE SLICES (round 6; the torch.distributed form of ffh_comm_set_exchange(1)): ONE all-to-all -- rank j receives
        every rank's records of slice j = [j * sl, (j + 1) * sl), sl = ceil(G / world) -- the slice folded locally in rank order (= database
        order: integer lanes add, overflow / cfd_max / jost_max take the maximum, the closest hit the minimum with its count summed at that
        level, the three f64 sums are added rank after rank), and ONE all-gather of the folded slices.  world x less payload into every
        rank than the three collectives over all G guides; the same bytes in self.summ afterwards."""
        torch, dist, G, W = self.torch, self.dist, self.G, self.world
        sl = (G + W - 1) // W
        if sl == 0:
            return self.summ
        send = torch.zeros(W * sl * self.itemsize, dtype=torch.uint8, device=self.device)
        send[: G * self.itemsize] = self.summ                        # slice j of the guides is chunk j; the tail pads the last slices
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)          # chunk r of recv = rank r's records of MY slice
        i32 = recv.view(torch.int32).view(W, sl, 22)
        f64 = recv.view(torch.float64).view(W, sl, 11)
        closest = i32[:, :, 8].to(torch.int64) & 0xFFFFFFFF           # 0xFFFFFFFF = none
        gmin = closest.min(0).values
        out = torch.zeros(sl * self.itemsize, dtype=torch.uint8, device=self.device)
        o32, o64 = out.view(torch.int32).view(sl, 22), out.view(torch.float64).view(sl, 11)
        cols = [0, 1, 3, 4, 5, 6, 7, 10, 11]                          # n_hits, ot_count, hist[5], in_genome, n_scored
        o32[:, cols] = i32[:, :, cols].sum(0, dtype=torch.int32)
        o32[:, 2] = i32[:, :, 2].max(0).values                        # overflow
        o32[:, 9] = torch.where(closest == gmin.unsqueeze(0), i32[:, :, 9], torch.zeros_like(i32[:, :, 9])).sum(0, dtype=torch.int32)
        o32[:, 8] = torch.where(gmin > 0x7FFFFFFF, gmin - (1 << 32), gmin).to(torch.int32)
        o64[:, 6] = f64[:, :, 6].max(0).values                        # cfd_max
        o64[:, 9] = f64[:, :, 9].max(0).values                        # jost_max
        acc = f64[0][:, [7, 8, 10]].clone()                           # cfd_sum, hsu_sum, jost_sum: rank after rank
        for r in range(1, W):
            acc += f64[r][:, [7, 8, 10]]
        o64[:, [7, 8, 10]] = acc
        gathered = torch.empty(W * sl * self.itemsize, dtype=torch.uint8, device=self.device)
        self._all_gather(gathered, out)
        self.summ.copy_(gathered[: G * self.itemsize])
        return self.summ

    def reduce_summaries_fused(self, ctx):
        """reduce_summaries with the packing, masking and unpacking done by three library kernels (the torch form costs ~20 small
        launches): same three collectives, same arithmetic"""
        dist, G = self.dist, self.G
        ctx.exchange_pack(self.summ.data_ptr(), G, self.mx.data_ptr(), self.sums.data_ptr(), self.fsum.data_ptr())
        dist.all_reduce(self.mx, op=dist.ReduceOp.MAX, group=self.group)
        ctx.exchange_mask(self.summ.data_ptr(), G, self.mx.data_ptr(), self.sums.data_ptr())
        dist.all_reduce(self.sums, op=dist.ReduceOp.SUM, group=self.group)
        self._all_gather(self.fsum_all, self.fsum)
        ctx.exchange_unpack(self.summ.data_ptr(), G, self.mx.data_ptr(), self.sums.data_ptr(), self.fsum_all.data_ptr(), self.world)
        return self.summ

    def step_two_pass(self, ctx, max_offtargets, jost=False):
        """the exchange in its first form: totals pass, all-gather, prior, full aggregation pass with the prior, reduction"""
        prior = self.prior_totals(ctx, max_offtargets)
        res = ctx.finalize_device_prior(max_offtargets, prior.data_ptr(), summaries_only=True, jost=jost)
        ctx.summaries_to_device(self.summ.data_ptr())
        self.reduce_summaries_fused(ctx)
        return res

    def step(self, ctx, max_offtargets, jost=False):
        """One aggregation pass per shard, stream-ordered, no host round trip: every shard aggregates as if it were the first
        (ffh_finalize_shard also yields its totals), the totals are all-gathered, and only the guides whose cut-off the earlier
        shards actually move -- prior > 0 and prior + shard total >= maximumOffTargets -- are aggregated again with the prior
        (ffh_finalize_shard_fixup).  Same reduced summaries as step_two_pass, bit for bit.  With device buffers the context is
        put on torch's current stream, so the library kernels and the RCCL collectives form one ordered sequence."""
        if self.summ.is_cuda and not getattr(ctx, "_on_caller_stream", False):
            ctx.use_stream(self.torch.cuda.current_stream().cuda_stream)
            ctx._on_caller_stream = True
        ctx.finalize_shard(max_offtargets, self.summ.data_ptr(), self.totals.data_ptr(), jost=jost)
        self._all_gather(self.all_totals, self.totals)
        ctx.exchange_prior(self.all_totals.data_ptr(), self.G, self.rank, max_offtargets, self.prior.data_ptr())
        ctx.finalize_shard_fixup(max_offtargets, self.prior.data_ptr(), self.totals.data_ptr(), self.summ.data_ptr(), jost=jost)
        self.reduce_summaries_fused(ctx)
        return None

    def summaries_numpy(self):
        """the reduced summaries on the host (through a page-locked staging tensor when the buffers live on a GPU)"""
        from . import capi
        if self.summ.is_cuda:
            if getattr(self, "_host", None) is None:
                self._host = self.torch.empty(self.summ.shape, dtype=self.torch.uint8, pin_memory=True)
            self._host.copy_(self.summ, non_blocking=True)
            self.torch.cuda.current_stream().synchronize()
            return self._host.numpy().view(capi.SUMMARY_DTYPE)
        return self.summ.numpy().view(capi.SUMMARY_DTYPE)


class MergedBulgeResult:
    """the per-shard results of ffh_discover_bulge merged in rank (= database) order: same fields as capi.BulgeResult"""

    def __init__(self, parts):
        n = parts[0]["guide_offsets"].shape[0] - 1
        counts = np.zeros(n, dtype=np.int64)
        for p in parts:
            counts += np.diff(p["guide_offsets"].astype(np.int64))
        self.n_guides = n
        self.guide_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        self.n_hits = int(self.guide_offsets[-1])
        fields = (("hit_targets", np.uint64), ("hit_mismatches", np.uint8), ("hit_bulge_type", np.uint8), ("hit_bulge_position", np.uint8))
        for f, dt in fields:
            setattr(self, f, np.zeros(self.n_hits, dtype=dt))
        fill = self.guide_offsets[:-1].astype(np.int64).copy()
        for p in parts:   # shard after shard: a guide's hits of shard r follow its hits of the shards before it
            off = p["guide_offsets"].astype(np.int64)
            cnt = np.diff(off)
            src = np.arange(int(off[-1]), dtype=np.int64)
            dst = np.repeat(fill - off[:-1], cnt) + src
            for f, _ in fields:
                getattr(self, f)[dst] = p[f]
            fill += cnt


def discover_bulge_sharded(ctx, guides, max_mismatch=3, max_bulge=1, tttv=False, group=None):
    """config C5 across bin shards (one rank per GPU): every rank runs ffh_discover_bulge on its resident shard -- there is no
    cut-off and no score in this search, so nothing is exchanged on the data path -- and the per-guide hit lists are concatenated in
    rank order, which is database order.  Returns the merged result on every rank."""
    import torch.distributed as dist
    res = ctx.discover_bulge(guides, max_mismatch, max_bulge, tttv=tttv)
    mine = {f: getattr(res, f) for f in ("guide_offsets", "hit_targets", "hit_mismatches", "hit_bulge_type", "hit_bulge_position")}
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return MergedBulgeResult([mine])
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    return MergedBulgeResult(parts)
