"""world_size-2/3 CPU tests (gloo) of the multi-GPU plumbing: shard bins, continue the ordered cut-off across ranks,
reduce the per-guide aggregates."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,max_ot", [(2, 40), (3, 2000), (2, 0)])
def test_sharded_discover_over_gloo(tmp_path, world, max_ot):
    out = str(tmp_path / "res.json")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, "5", str(max_ot)]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.load(open(out))
    assert res["world"] == world
    for k in ("ok_hits", "ok_totals", "ok_overflow", "ok_hist", "ok_closest", "ok_device_exchange", "ok_sliced_exchange"):
        assert res[k], (k, res)
    assert res["max_cfd_err"] <= 1e-9 and res["max_cfdmax_err"] == 0.0 and res["max_hsu_err"] <= 1e-9 and res["max_jost_err"] <= 1e-9, res
    if max_ot == 40:
        assert 0 < res["n_overflowed"] < res["n_guides"] and res["crossing"] > 0  # the cut-off really crossed a shard boundary


def test_shard_bins_balances_bytes():
    from flashfry_amd import dist as ffdist
    rng = np.random.default_rng(0)
    sizes = rng.integers(8, 10000, size=16384)
    for world in (1, 2, 3, 8):
        cuts = ffdist.shard_bins(sizes, world)
        assert cuts[0][0] == 0 and cuts[-1][1] == 16384
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        loads = [sizes[a:b].sum() for a, b in cuts]
        assert max(loads) - min(loads) <= 2 * sizes.max()
    assert ffdist.shard_bins([0, 0, 0, 0], 2) == [(0, 0), (0, 4)] or len(ffdist.shard_bins([0, 0, 0, 0], 2)) == 2


def test_bulge_results_of_shards_merge_in_rank_order():
    """flashfry_amd.dist.MergedBulgeResult: per guide, the hits of shard 0, then of shard 1, ... (database order)"""
    import numpy as np
    from flashfry_amd.dist import MergedBulgeResult
    rng = np.random.default_rng(3)
    parts, per_guide = [], [[] for _ in range(7)]
    for shard in range(3):
        cnt = rng.integers(0, 5, size=7)
        off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
        n = int(off[-1])
        p = {"guide_offsets": off, "hit_targets": rng.integers(0, 1 << 40, size=n, dtype=np.uint64), "hit_mismatches": rng.integers(0, 4, size=n).astype(np.uint8),
             "hit_bulge_type": rng.integers(0, 3, size=n).astype(np.uint8), "hit_bulge_position": rng.integers(0, 19, size=n).astype(np.uint8)}
        parts.append(p)
        for g in range(7):
            for h in range(int(off[g]), int(off[g + 1])):
                per_guide[g].append((int(p["hit_targets"][h]), int(p["hit_mismatches"][h]), int(p["hit_bulge_type"][h]), int(p["hit_bulge_position"][h])))
    m = MergedBulgeResult(parts)
    assert m.n_hits == sum(len(x) for x in per_guide)
    for g in range(7):
        a, b = int(m.guide_offsets[g]), int(m.guide_offsets[g + 1])
        got = list(zip(m.hit_targets[a:b].tolist(), m.hit_mismatches[a:b].tolist(), m.hit_bulge_type[a:b].tolist(), m.hit_bulge_position[a:b].tolist()))
        assert got == per_guide[g]


def test_exchange_kernels_emulated_on_the_cpu(tmp_path):
    """round 6: the kernels of the shards' exchange, compiled by g++ from the source the GPU build compiles and run thread after thread
    (tests/exchange_emul_main.cpp): the exchange by guide slices -- pack, all-to-all, fold per slice, priors and flag back, assemble,
    all-gather of the folded slices -- leaves on every shard the prior, the flag word and the reduced records of the all-gather form;
    world 2 / 3 / 5 / 8, guide counts the world size does not divide, fewer guides than shards, failing shards, both rounds."""
    exe = str(tmp_path / "exchange_emul")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-fno-strict-aliasing", "-Wno-unknown-pragmas", "-o", exe, os.path.join(ROOT, "tests", "exchange_emul_main.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "leaves what the all-gather form leaves" in r.stdout, r.stdout[-2000:]
