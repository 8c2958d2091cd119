
"""the repeat-structured workload of bench.py alone: unbounded, then bounded; every field of ffh_timings (dev tool, GPU box)"""
import sys, time, json, numpy as np, torch
sys.path.insert(0, ".")
from flashfry_amd import capi, synth
dev = torch.device("cuda:0")
db = synth.make_repeat_database(int(3.0e8), seed=synth.DB_SEED + 99, device=dev)
guides = synth.make_guides_from_database(db, 100000, device=dev).cpu().numpy().view(np.uint64)
with capi.Context(3) as ctx:
    torch.cuda.synchronize()
    ctx.load_soa_device(db["targets"].data_ptr(), db["T"], db["positions"].data_ptr(), db["P"])
    del db; torch.cuda.empty_cache()
    for mode in (0, 1):
        ctx.set_bounding(mode)
        res = ctx.discover(guides, 4, 2000, summaries_only=True)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = ctx.discover(guides, 4, 2000, summaries_only=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        tm = ctx.timings().as_dict()
        print("bounding", mode, "ms", round(float(np.median(ts)), 2), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in tm.items() }, "digest", hash(res.summaries.tobytes()) & 0xffffff)
