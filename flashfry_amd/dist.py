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
"""
import numpy as np


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
