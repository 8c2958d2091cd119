#!/usr/bin/env python3
"""Where the sharded step's extra time goes (1-rank RCCL group on one GPU): stage timings of DeviceExchange.step."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from flashfry_amd import capi, dist as ffdist, synth  # noqa: E402

dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
G, T = 100000, int(float(sys.argv[1]) if len(sys.argv) > 1 else 3e8)
guides_dev = synth.make_guides(G, device=dev)
db = synth.make_database(T, plant_guides=guides_dev, device=dev)
g = guides_dev.cpu().numpy().view(np.uint64)
ctx = capi.Context(3)
torch.cuda.synchronize()
ctx.load_soa_device(db["targets"].data_ptr(), db["T"], db["positions"].data_ptr(), db["P"])
del db
ex = ffdist.DeviceExchange(G, dev)
acc = {}


def lap(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return r


N = 6
for it in range(N + 1):
    if it == 1:
        acc.clear()
    lap("scan", lambda: ctx.scan(g, 4))
    prior = lap("prior_totals", lambda: ex.prior_totals(ctx, 2000))
    lap("finalize", lambda: ctx.finalize_device_prior(2000, prior.data_ptr()))
    lap("summaries_to_device", lambda: ctx.summaries_to_device(ex.summ.data_ptr()))
    lap("reduce", lambda: ex.reduce_summaries_fused(ctx))
    lap("to_host", lambda: ex.summaries_numpy())
print({k: round(v / N, 3) for k, v in acc.items()})
ctx.close()
dist.destroy_process_group()
