#!/usr/bin/env python3
"""one stress case (kind 0) run again and again against the oracle's answer: a rare difference is a race or a stale buffer (dev tool, GPU box)
  python tools/debug_loop.py seed n_targets n_guides max_mm max_ot bounding iterations [fresh]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from flashfry_amd import capi
import oracle_lib
from helpers import make_case
seed, T, G, mm, max_ot, bounding, iters = [int(x) for x in sys.argv[1:8]]
fresh = len(sys.argv) > 8
oracle = oracle_lib.load()
odb, t, p, g = make_case(oracle, T, G, enzyme=3, seed=seed)
ora = odb.discover(g, mm, max_ot)
idx = None
fails = 0
ctx = None
for it in range(iters):
    if ctx is None or fresh:
        if ctx is not None:
            ctx.close()
        ctx = capi.Context(3)
        ctx.load_soa(t, p)
        ctx.set_bounding(bounding)
    gpu = ctx.discover(g, mm, max_ot, jost=True)
    if not np.array_equal(gpu.guide_offsets, ora.guide_offsets) or not np.array_equal(gpu.hit_targets, ora.hit_targets):
        fails += 1
        bad = [k for k in range(len(g)) if not np.array_equal(gpu.hits(k), ora.hits(k))]
        if idx is None:
            idx = {int(v): i for i, v in enumerate(t)}
        k = bad[0]
        a, b = set(int(x) for x in gpu.hits(k)), set(int(x) for x in ora.hits(k))
        tm = ctx.timings().as_dict()
        print("iteration %d: %d guides differ %s; guide %d: gpu %d oracle %d missing %s extra %s; raw %d slabs %d" % (
            it, len(bad), bad[:6], k, len(a), len(b), sorted(idx[v] for v in b - a)[:6], sorted(idx[v] for v in a - b)[:6], tm["n_raw_hits"], tm["bounded_slabs"]), flush=True)
print("env %s fresh %s: %d failures in %d iterations" % ({k: v for k, v in os.environ.items() if k.startswith("FFH_")}, fresh, fails, iters))
