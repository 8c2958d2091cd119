#!/usr/bin/env python3
"""Randomised parity sweep against the oracle (dev tool, GPU box): many seeds x sizes x mismatch budgets x cut-offs through the
same helpers as tests/test_gpu_parity.py.  Prints one line per case and stops at the first difference.
  python tools/stress_parity.py [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flashfry_amd import capi
import oracle_lib
from helpers import make_case, assert_same_hits, assert_same_scores
from test_gpu_parity import dense_case

oracle = oracle_lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(12345)
t0, n = time.time(), 0
while time.time() - t0 < budget:
    seed = int(rng.integers(0, 1 << 30))
    kind = int(rng.integers(0, 2))
    max_mm = int(rng.choice([0, 1, 2, 3, 4, 4, 4, 5, 6]))
    max_ot = int(rng.choice([5, 40, 60, 300, 2000]))
    if kind == 0:
        odb, t, p, g = make_case(oracle, int(rng.integers(100, 400000)), int(rng.integers(1, 600)), enzyme=3, seed=seed)
    else:
        ng = int(rng.integers(10, 500))
        odb, t, p, g = dense_case(oracle, n_random=int(rng.integers(1000, 120000)), n_guides=ng,
                                  n_dense=int(rng.integers(1, min(60, ng))), variants=int(rng.integers(10, 200)), seed=seed)
    enz = 3
    with capi.Context(enz) as ctx:
        ctx.load_soa(t, p)
        gpu = ctx.discover(g, max_mm, max_ot, jost=True)
        only = ctx.finalize(max_ot, summaries_only=True, jost=True)
        assert only.summaries.tobytes() == gpu.summaries.tobytes()
    ora = odb.discover(g, max_mm, max_ot)
    assert_same_hits(gpu, ora)
    assert_same_scores(oracle, enz, g, gpu, ora)
    n += 1
    print("ok %3d kind %d enzyme %d T %7d G %4d mm %d max_ot %4d hits %8d" % (n, kind, enz, len(t), len(g), max_mm, max_ot, gpu.n_hits), flush=True)
print("all %d cases agree" % n)
