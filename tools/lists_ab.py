#!/usr/bin/env python3
"""list-delivering ffh_discover at hg38 scale: default table (no per-hit scores, no positions) and with positions; median of 7 calls.
FFH_PIPELINE=0 in the environment keeps the guide set in one piece (A/B of the two-part pipeline)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flashfry_amd import capi, synth
dev = torch.device("cuda:0")
gd = synth.make_guides(100000, device=dev)
db = synth.make_database(int(3.0e8), seed=synth.DB_SEED, plant_guides=gd, device=dev)
g = gd.cpu().numpy().view(np.uint64)
out = {"pipeline": os.environ.get("FFH_PIPELINE", "1")}
with capi.Context(3) as ctx:
    torch.cuda.synchronize()
    ctx.load_soa_device(db["targets"].data_ptr(), db["T"], db["positions"].data_ptr(), db["P"])
    for name, kw in (("default_table", dict(positions=False, hit_scores=False)), ("with_positions", dict(hit_scores=False)), ("everything", dict())):
        ts = []
        for _ in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = ctx.discover_device(gd.data_ptr(), 100000, 4, 2000, **kw)
            ts.append((time.perf_counter() - t0) * 1e3)
        out[name] = round(float(np.median(ts[2:])), 3)
        out[name + "_digest"] = int(np.bitwise_xor.reduce(r.hit_targets)) ^ int(r.guide_offsets.sum()) ^ (int(r.positions.sum()) if r.positions is not None else 0)
print(json.dumps(out))
