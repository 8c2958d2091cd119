#!/usr/bin/env python3
"""Does a device-to-host copy overlap the scan on this stack?  The aggregates-only step at hg38 scale (a) alone, (b) with a 104 MB
device-to-pinned-host copy running on another stream meanwhile, (c) the copy alone; also with 50 000 guides (half a guide set).
Decides whether delivering half A's hit lists under half B's scan can pay."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from flashfry_amd import capi, synth
    dev = torch.device("cuda:0")
    G = 100000
    gd = synth.make_guides(G, device=dev)
    db = synth.make_database(int(3.0e8), seed=synth.DB_SEED, plant_guides=gd, device=dev)
    src = torch.empty(104_000_000 // 8, dtype=torch.int64, device=dev).fill_(7)
    dst = torch.empty(104_000_000 // 8, dtype=torch.int64, pin_memory=True)
    side = torch.cuda.Stream()
    out = {}
    with capi.Context(3) as ctx:
        torch.cuda.synchronize()
        ctx.load_soa_device(db["targets"].data_ptr(), db["T"], db["positions"].data_ptr(), db["P"])
        for ng in (G, G // 2):
            def step():
                ctx.scan_device(gd.data_ptr(), ng, 4)
                return ctx.finalize(2000, summaries_only=True)
            for _ in range(4):
                step()
            def timed(with_copy, with_step):
                ts = []
                for _ in range(10):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    if with_copy:
                        with torch.cuda.stream(side):
                            dst.copy_(src, non_blocking=True)
                    if with_step:
                        step()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                return float(np.median(ts))
            out["guides_%d" % ng] = {"step_alone_ms": timed(False, True), "copy_alone_ms": timed(True, False), "both_ms": timed(True, True),
                                     "compare_ms_with_copy": None}
            timed(True, True)
            out["guides_%d" % ng]["compare_ms_with_copy"] = ctx.timings().compare_ms
    print(json.dumps(out))


if __name__ == "__main__":
    main()
