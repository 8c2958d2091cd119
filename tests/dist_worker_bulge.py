"""Worker of tests/test_dist_gpu.py: config C5 (Cas12a, mismatches + one bulge) over bin shards.  Every rank loads its contiguous bin
range of one TTTN database into its own context on the GPU, runs the seeded bulge search on it, and
flashfry_amd.dist.discover_bulge_sharded concatenates the per-guide hit lists in rank order; rank 0 compares with the unsharded
search of the whole database."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from flashfry_amd import capi, dist as ffdist  # noqa: E402
from tests.test_gpu_parity import _cas12a_database  # noqa: E402


def main():
    out_path = sys.argv[1]
    n_dev = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(n_dev, 1)
    torch.cuda.set_device(local)
    dist.init_process_group(os.environ.get("FFH_TEST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(5)   # the same database on every rank
    guides40 = rng.integers(0, 1 << 40, size=500, dtype=np.uint64)
    targets, positions = _cas12a_database(rng, 400_000, guides40[:200], 15)
    guides = guides40 | np.uint64(0b11111100 << 40) | np.uint64(1 << 48)
    binidx = ((targets >> np.uint64(26)) & np.uint64(0x3FFF)).astype(np.int64)   # the 7 bases after the 5' PAM (BinWriter.scala:58-64)
    sizes = np.bincount(binidx, minlength=1 << 14) * 16
    b0, b1 = ffdist.shard_bins(sizes, world)[rank]
    lo, hi = int(np.searchsorted(binidx, b0, side="left")), int(np.searchsorted(binidx, b1, side="left"))
    with capi.Context(1, device=local) as ctx:
        ctx.load_soa(targets[lo:hi], positions[lo:hi])
        merged = ffdist.discover_bulge_sharded(ctx, guides, 3, 1, tttv=True)
    if rank == 0:
        with capi.Context(1, device=local) as full_ctx:
            full_ctx.load_soa(targets, positions)
            full = full_ctx.discover_bulge(guides, 3, 1, tttv=True)
        fields = ("guide_offsets", "hit_targets", "hit_mismatches", "hit_bulge_type", "hit_bulge_position")
        res = {"world": world, "n_hits": int(full.n_hits), "shard_targets": hi - lo, "types": sorted(set(int(x) for x in full.hit_bulge_type)),
               "ok": all(bool(np.array_equal(getattr(merged, f), getattr(full, f))) for f in fields)}
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()
    os._exit(0)


if __name__ == "__main__":
    main()
