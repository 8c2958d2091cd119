"""Worker of tests/test_dist_cpu.py: one process per database shard, gloo backend (no GPU).  The per-shard hit lists
come from the CPU oracle (test infrastructure) -- what is under test is flashfry_amd.dist: the ordered cut-off prefix
over ranks and the reduction of the per-guide aggregates."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from flashfry_amd import capi, dist as ffdist, synth  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.test_gpu_parity import dense_case  # noqa: E402


def main():
    out_path, max_mm, max_ot = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    oracle = oracle_lib.load()
    odb, targets, positions, guides = dense_case(oracle, seed=31)
    # contiguous bin shards balanced by payload bytes
    sizes = [len(odb.bin(b)[0]) * 8 for b in range(odb.n_bins)]
    b0, b1 = ffdist.shard_bins(sizes, world)[rank]
    binidx = ((targets >> np.uint64(32)) & np.uint64(0x3FFF)).astype(np.int64)
    sel = (binidx >= b0) & (binidx < b1)
    poff = np.concatenate([[0], np.cumsum(targets >> np.uint64(48))]).astype(np.int64)
    lo, hi = int(np.argmax(sel)) if sel.any() else 0, (len(sel) - int(np.argmax(sel[::-1]))) if sel.any() else 0
    shard = oracle.db_from_sorted(3, targets[lo:hi], positions[poff[lo]:poff[hi]], contigs=synth.CONTIGS_24)
    raw = shard.discover(guides, max_mm, 2 ** 30)  # every hit of this shard, database order
    G = len(guides)
    counts = [(raw.hits(g) >> np.uint64(48)).astype(np.int64) for g in range(G)]
    totals = np.array([min(int(c.sum()), max_ot) for c in counts], dtype=np.uint32)
    prior = ffdist.prior_totals(totals, max_ot)
    summ = np.zeros(G, dtype=capi.SUMMARY_DTYPE)
    kept = []
    for g in range(G):
        run, k = int(prior[g]), 0
        for c in counts[g]:  # CRISPRSiteOT.addOT/full continued from the lower-ranked shards
            if run >= max_ot:
                break
            run += int(c)
            k += 1
        hits = raw.hits(g)[:k]
        kept.append([int(x) for x in hits])
        s, per = oracle.score_guide(3, int(guides[g]), hits)
        m = summ[g]
        m["n_hits"], m["ot_count"], m["overflow"] = k, run - int(prior[g]), int(run >= max_ot)
        m["hist"] = list(s.hist)
        m["closest"] = 0xFFFFFFFF if s.closest == 2 ** 31 - 1 else s.closest
        m["closest_count"], m["in_genome"] = s.closest_count, s.in_genome
        scored = per[~np.isnan(per)]
        m["n_scored"] = len(scored)
        m["cfd_max"] = float(scored.max()) if len(scored) else 0.0
        m["cfd_sum"] = float(1.0 / s.cfd_spec - 1.0) if len(scored) else 0.0
        m["hsu_sum"] = float(100.0 * 100.0 / s.hsu - 100.0)
        m["jost_max"], m["jost_sum"] = s.jost_max, float(1.0 / s.jost_spec - 1.0)
    # the device-resident exchange used by bench.py (here on CPU tensors over gloo) must agree with the host path bit for bit
    import torch
    ex = ffdist.DeviceExchange(G, "cpu")
    ex.totals.copy_(torch.from_numpy(totals.astype(np.int32)))
    prior_dev = ex.prior_from_totals(max_ot).numpy().astype(np.uint32)
    ex.summ.copy_(torch.from_numpy(summ.view(np.uint8).reshape(-1).copy()))
    ex.reduce_summaries()
    # ... and so must the exchange by guide slices (one all-to-all + one all-gather: ffh_comm_set_exchange(1)'s torch.distributed form)
    ex2 = ffdist.DeviceExchange(G, "cpu")
    ex2.summ.copy_(torch.from_numpy(summ.view(np.uint8).reshape(-1).copy()))
    ex2.reduce_summaries_sliced()
    ffdist.allreduce_summaries(summ)
    same_exchange = bool(np.array_equal(prior_dev, prior)) and ex.summ.numpy().tobytes() == summ.tobytes()
    same_sliced = ex2.summ.numpy().tobytes() == summ.tobytes()
    gathered = [None] * world
    dist.all_gather_object(gathered, kept)
    if rank == 0:
        full = odb.discover(guides, max_mm, max_ot)
        ok_hits = all(sum((gathered[r][g] for r in range(world)), []) == [int(x) for x in full.hits(g)] for g in range(G))
        exp = [oracle.score_guide(3, int(guides[g]), full.hits(g))[0] for g in range(G)]
        res = {
            "world": world, "ok_hits": bool(ok_hits), "ok_device_exchange": same_exchange, "ok_sliced_exchange": same_sliced,
            "ok_totals": bool(np.array_equal(summ["ot_count"].astype(np.int64), full.current_total)),
            "ok_overflow": bool(np.array_equal(summ["overflow"].astype(bool), full.full)),
            "ok_hist": all(list(summ["hist"][g]) == list(exp[g].hist) for g in range(G)),
            "ok_closest": all(int(summ["closest"][g]) == (0xFFFFFFFF if exp[g].closest == 2 ** 31 - 1 else exp[g].closest)
                              and int(summ["closest_count"][g]) == exp[g].closest_count for g in range(G)),
            "max_cfd_err": float(max(abs(1.0 / (1.0 + summ["cfd_sum"][g]) - exp[g].cfd_spec) for g in range(G))),
            "max_cfdmax_err": float(max(abs(summ["cfd_max"][g] - exp[g].cfd_max) for g in range(G))),
            "max_jost_err": float(max(max(abs(summ["jost_max"][g] - exp[g].jost_max), abs(1.0 / (1.0 + summ["jost_sum"][g]) - exp[g].jost_spec)) for g in range(G))),
            "max_hsu_err": float(max(abs(100.0 / (100.0 + summ["hsu_sum"][g]) * 100.0 - exp[g].hsu) for g in range(G))),
            "n_overflowed": int(full.full.sum()), "n_guides": G,
            "crossing": int(sum(1 for g in range(G) if 0 < len(gathered[0][g]) and 0 < sum(len(gathered[r][g]) for r in range(1, world)))),
        }
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
