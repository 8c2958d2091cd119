#!/usr/bin/env python3
"""Where does the host-side wall time of one discover step go?  (dev tool; prints a breakdown)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flashfry_amd import capi, synth

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300000000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dev = torch.device("cuda", 0)
g = synth.make_guides(G, device=dev)
db = synth.make_database(T, plant_guides=g, device=dev)
gn = g.cpu().numpy().view(np.uint64)
ctx = capi.Context(3)
torch.cuda.synchronize()
ctx.load_soa_device(db["targets"].data_ptr(), db["T"], db["positions"].data_ptr(), db["P"])
del db
for it in range(4):
    t0 = time.perf_counter(); ctx.scan(gn, 4); t1 = time.perf_counter()
    r = ctx.finalize(2000, summaries_only=True); t2 = time.perf_counter()
    tm = ctx.timings()
    print("iter %d: scan wall %.2f ms (device %.2f: prep %.2f cmp %.2f sort %.2f)  finalize wall %.2f ms (device %.2f)" % (
        it, (t1 - t0) * 1e3, tm.total_scan_ms, tm.prepare_ms, tm.compare_ms, tm.sort_ms, (t2 - t1) * 1e3, tm.finalize_ms))
