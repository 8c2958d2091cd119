"""Worker of tests/test_dist_gpu.py: TWO RANKS WITH REAL HIP SHARDS.  Every rank loads its contiguous bin shard of one database into
its own context on the GPU (the test box has one: both ranks share it; on a multi-GPU node LOCAL_RANK picks the device), scans it
with the HIP kernels, and the ranks run the device-resident exchange of bench.py (flashfry_amd.dist.DeviceExchange.step:
ffh_finalize_shard -> all-gather of the totals -> ffh_exchange_prior -> ffh_finalize_shard_fixup -> the three reduction collectives)
over torch.distributed.  Rank 0 compares with the UNSHARDED discover of the same database on the same GPU: the reduced aggregates,
and the per-rank hit lists (finalized with the prior totals) concatenated in rank order."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from flashfry_amd import capi, dist as ffdist  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.test_gpu_parity import dense_case  # noqa: E402


def main():
    out_path, max_mm, max_ot = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    backend = os.environ.get("FFH_TEST_BACKEND", "gloo")
    n_dev = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(n_dev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    rank, world = dist.get_rank(), dist.get_world_size()
    oracle = oracle_lib.load()   # only to build the synthetic case (dense neighbourhoods around some guides)
    odb, targets, positions, guides = dense_case(oracle, n_random=200000, n_guides=300, n_dense=40, variants=120, seed=31)
    G = len(guides)
    # contiguous bin shards balanced by payload bytes (BinaryHeader's uncompressedSize)
    sizes = [len(odb.bin(b)[0]) * 8 for b in range(odb.n_bins)]
    b0, b1 = ffdist.shard_bins(sizes, world)[rank]
    binidx = ((targets >> np.uint64(32)) & np.uint64(0x3FFF)).astype(np.int64)
    lo, hi = int(np.searchsorted(binidx, b0, side="left")), int(np.searchsorted(binidx, b1, side="left"))
    poff = np.concatenate([[0], np.cumsum(targets >> np.uint64(48))]).astype(np.int64)
    with capi.Context(3, device=local) as ctx:
        ctx.load_soa(targets[lo:hi], positions[poff[lo]:poff[hi]])
        ctx.scan(guides, max_mm)
        ex = ffdist.DeviceExchange(G, dev)
        ex.step(ctx, max_ot, jost=True)                  # the path bench.py --gpus N times
        reduced = ex.summaries_numpy().copy()
        ex2 = ffdist.DeviceExchange(G, dev)
        ex2.step_two_pass(ctx, max_ot, jost=True)        # totals pass + full pass with the prior: must agree bit for bit
        same_two_pass = ex2.summaries_numpy().tobytes() == reduced.tobytes()
        # the hit lists of this shard under the ordered cut-off continued from the lower ranks
        lists = ctx.finalize(max_ot, prior_totals=ex.prior.cpu().numpy().astype(np.uint32))
        kept = [[int(x) for x in lists.hits(g)] for g in range(G)]
    gathered = [None] * world
    dist.all_gather_object(gathered, kept)
    if rank == 0:
        with capi.Context(3, device=local) as full_ctx:
            full_ctx.load_soa(targets, positions)
            full = full_ctx.discover(guides, max_mm, max_ot, jost=True)
        s, r = full.summaries, reduced
        ints = ("n_hits", "ot_count", "overflow", "hist", "closest", "closest_count", "in_genome", "n_scored")
        res = {
            "world": world, "ok_two_pass": bool(same_two_pass),
            "ok_hits": all(sum((gathered[k][g] for k in range(world)), []) == [int(x) for x in full.hits(g)] for g in range(G)),
            "ok_ints": all(bool(np.array_equal(s[f], r[f])) for f in ints),
            "ok_max": bool(np.array_equal(s["cfd_max"], r["cfd_max"]) and np.array_equal(s["jost_max"], r["jost_max"])),
            # the f64 sums are added shard by shard in rank order: same value up to the association of the additions
            "max_sum_err": float(max(np.abs(s[f] - r[f]).max() for f in ("cfd_sum", "hsu_sum", "jost_sum"))),
            "n_overflowed": int(s["overflow"].sum()), "n_guides": G,
            "crossing": int(sum(1 for g in range(G) if 0 < len(gathered[0][g]) and 0 < sum(len(gathered[k][g]) for k in range(1, world)))),
            "cut_in_later_shard": int(sum(1 for g in range(G) if s["overflow"][g] and sum(len(gathered[k][g]) for k in range(1, world)) > 0)),
        }
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()
    os._exit(0)


if __name__ == "__main__":
    main()
