#!/usr/bin/env python3
"""What one rank of an N-way strong-scaling run does, on one GPU: the aggregates-only step against shard `rank` of N contiguous
shards (by targets) of the hg38-scale synthetic database.  Prints the step time and the breakdown."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--targets", type=float, default=3.0e8)
    ap.add_argument("--guides", type=int, default=100000)
    ap.add_argument("--plan-a", type=int, default=-1, help="force the prefix width (default: the library's choice)")
    ap.add_argument("--plan-r1", type=int, default=-1, help="force the prefix radius")
    ap.add_argument("--comm", action="store_true", help="time ffh_discover_sharded through a one-rank RCCL communicator on the shard: the sharded run's own code path (device "
                    "summaries, the all-gather, the local fold, rank 0's copy-out) -- what one rank does, minus the links")
    args = ap.parse_args()
    import torch
    from flashfry_amd import capi, synth
    dev = torch.device("cuda:0")
    gd = synth.make_guides(args.guides, device=dev)
    db = synth.make_database(int(args.targets), seed=synth.DB_SEED, plant_guides=gd, device=dev)
    T = db["T"]
    lo, hi = T * args.rank // args.shards, T * (args.rank + 1) // args.shards
    plo, phi = int(db["pos_offsets"][lo]), int(db["pos_offsets"][hi])
    t, p = db["targets"][lo:hi].contiguous(), db["positions"][plo:phi].contiguous()
    with capi.Context(3) as ctx:
        torch.cuda.synchronize()
        if args.plan_a >= 0 or args.plan_r1 >= 0:
            ctx.set_plan(args.plan_a, args.plan_r1)
        ctx.load_soa_device(t.data_ptr(), hi - lo, p.data_ptr(), phi - plo)
        for _ in range(3):
            ctx.scan_device(gd.data_ptr(), args.guides, 4)
            ctx.finalize(2000, summaries_only=True)
        ts, tms = [], []
        for _ in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.scan_device(gd.data_ptr(), args.guides, 4)
            res = ctx.finalize(2000, summaries_only=True)
            ts.append((time.perf_counter() - t0) * 1e3)
            tms.append(ctx.timings().as_dict())
        if args.comm:
            with capi.Comm.rank(ctx, 0, 1, capi.comm_unique_id()) as comm:
                out = capi.host_summaries(args.guides)
                for _ in range(3):
                    comm.discover_device(gd.data_ptr(), args.guides, 4, 2000, out=out)
                tc, ex = [], []
                for _ in range(10):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    comm.discover_device(gd.data_ptr(), args.guides, 4, 2000, out=out)
                    tc.append((time.perf_counter() - t0) * 1e3)
                    ex.append(comm.timings())
                assert out.tobytes() == res.summaries.tobytes()
                print(json.dumps({"shards": args.shards, "rank": args.rank, "comm_one_rank_ms_per_step": float(np.median(tc)),
                                  "scan_ms": float(np.median([e["scan_ms"] for e in ex])), "exchange_ms": float(np.median([e["exchange_ms"] for e in ex]))}))
        print(json.dumps({"shards": args.shards, "rank": args.rank, "targets": hi - lo, "ms_per_step": float(np.median(ts)),
                          "breakdown_ms": {k: round(float(np.mean([x[k] for x in tms])), 3) for k in ("prepare_ms", "compare_ms", "sort_ms", "finalize_ms", "total_scan_ms")},
                          "plan": [int(tms[-1]["prefix_bases"]), int(tms[-1]["prefix_radius"]), int(tms[-1]["suffix_radius"])], "tiles_prefix": int(tms[-1]["tiles_prefix"]), "pairs": int(tms[-1]["pairs_prefix"] + tms[-1]["pairs_suffix"]), "hits": int(res.summaries["n_hits"].sum())}))


if __name__ == "__main__":
    main()
