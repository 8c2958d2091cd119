#!/usr/bin/env python3
"""The reference's published timing grid on one MI355X: guide counts {1 .. 100 000} x maxMismatch {3, 4, 5} against an
hg38-scale database (paper/run_timing_collection.py:3-4 loops the same grid; BASELINE.md section 1 holds its results).

Each cell is a complete `discover` through the C ABI: scan + ordered cut-off + CFD/Hsu2013 aggregates + the retained hit
lists and their positions copied to the host (what `ffh_discover` returns), median of --repeats calls, database already
resident. Prints one JSON object and a Markdown table; nothing here reads the oracle or the reference.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PUBLISHED = {  # single-core JVM wall time, seconds (min of the replicate sets)
    (1, 4): 11, (10, 4): 34, (100, 4): 41, (1000, 3): 44, (1000, 4): 61, (1000, 5): 104,
    (10000, 3): 82, (10000, 4): 203, (10000, 5): 605, (100000, 3): 497, (100000, 4): 1818, (100000, 5): 7537,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=float, default=3.0e8)
    ap.add_argument("--guides", default="1,10,100,1000,10000,100000")
    ap.add_argument("--mismatches", default="3,4,5")
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--max-offtargets", type=int, default=2000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "timing_grid.json"))
    args = ap.parse_args()
    import torch
    from flashfry_amd import capi, synth

    dev = torch.device("cuda", 0)
    gmax = max(int(g) for g in args.guides.split(","))
    guides_dev = synth.make_guides(gmax, device=dev)
    db = synth.make_database(int(args.targets), seed=synth.DB_SEED, plant_guides=guides_dev, device=dev)
    T, P = db["T"], db["P"]
    guides_np = guides_dev.cpu().numpy().view(np.uint64)
    ctx = capi.Context(3, device=0)
    torch.cuda.synchronize()
    ctx.load_soa_device(db["targets"].data_ptr(), T, db["positions"].data_ptr(), P)
    del db
    torch.cuda.empty_cache()
    cells = []
    for mm in [int(m) for m in args.mismatches.split(",")]:
        for G in [int(g) for g in args.guides.split(",")]:
            g = np.ascontiguousarray(guides_np[:G])
            walls, tm = [], None
            for r in range(args.repeats + 1):  # first call warms the per-call buffers
                t0 = time.perf_counter()
                res = ctx.discover(g, mm, args.max_offtargets)
                dt = time.perf_counter() - t0
                if r:
                    walls.append(dt)
                tm = ctx.timings().as_dict()
                kept_hits, kept_pos = int(res.n_hits), int(res.summaries["ot_count"].sum())
                over = int(res.summaries["overflow"].sum())
                del res
            # what the reference's `discover` delivers: the default table (sequence_count_mismatches per hit: no positions, no per-hit
            # scores; modules/OffTargetDiscovery.scala:51-53) and the table with --positionOutput
            forms = {}
            for name, kw in (("default_table", dict(positions=False, hit_scores=False)), ("with_positions", dict(hit_scores=False))):
                ws = []
                for r in range(args.repeats + 1):
                    t0 = time.perf_counter()
                    res = ctx.discover(g, mm, args.max_offtargets, **kw)
                    dt = time.perf_counter() - t0
                    if r:
                        ws.append(dt)
                    del res
                forms[name] = float(np.median(ws)) * 1e3
            cell = {"guides": G, "max_mismatch": mm, "wall_ms": float(np.median(walls)) * 1e3, "default_table_ms": forms["default_table"], "with_positions_ms": forms["with_positions"],
                    "compare_ms": tm["compare_ms"], "prepare_ms": tm["prepare_ms"],
                    "sort_ms": tm["sort_ms"], "finalize_ms": tm["finalize_ms"], "raw_hits": tm["n_raw_hits"], "kept_hits": kept_hits, "kept_positions": kept_pos,
                    "overflowed_guides": over, "executed_comparisons": tm["pairs_prefix"] + tm["pairs_suffix"], "launches": tm["compare_launches"],
                    "plan": [tm["prefix_bases"], tm["prefix_radius"], tm["suffix_radius"]], "published_jvm_1core_s": PUBLISHED.get((G, mm))}
            cells.append(cell)
            print(json.dumps(cell), flush=True)
    out = {"targets": T, "positions": P, "max_offtargets": args.max_offtargets, "repeats": args.repeats, "db_prepare_ms": ctx.info().prepare_ms, "cells": cells}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("| guides | mm | discover: default table (ms) | with positions (ms) | + per-hit scores (ms) | compare (ms) | raw hits | kept positions | overflowed | published JVM 1 core (s) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for c in cells:
        print("| %d | %d | %.2f | %.2f | %.2f | %.2f | %d | %d | %d | %s |" % (c["guides"], c["max_mismatch"], c["default_table_ms"], c["with_positions_ms"], c["wall_ms"], c["compare_ms"],
                                                                             c["raw_hits"], c["kept_positions"], c["overflowed_guides"], c["published_jvm_1core_s"] or "-"))
    ctx.close()


if __name__ == "__main__":
    main()
