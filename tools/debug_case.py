#!/usr/bin/env python3
"""replay one stress case (kind 0) and report which guides differ from the oracle (dev tool, GPU box)
  python tools/debug_case.py seed n_targets n_guides max_mm max_ot bounding"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from flashfry_amd import capi
import oracle_lib
from helpers import make_case
seed, T, G, mm, max_ot, bounding = [int(x) for x in sys.argv[1:7]]
oracle = oracle_lib.load()
odb, t, p, g = make_case(oracle, T, G, enzyme=3, seed=seed)
ora = odb.discover(g, mm, max_ot)
with capi.Context(3) as ctx:
    ctx.load_soa(t, p)
    ctx.set_bounding(bounding)
    gpu = ctx.discover(g, mm, max_ot)
    tm = ctx.timings().as_dict()
bad = [k for k in range(len(g)) if not np.array_equal(gpu.hits(k), ora.hits(k))]
print("env", {k: v for k, v in os.environ.items() if k.startswith("FFH_")}, "plan", tm["prefix_bases"], tm["prefix_radius"], tm["suffix_radius"], "slabs", tm["bounded_slabs"], "raw", tm["n_raw_hits"],
      "bad guides", len(bad), bad[:10])
for k in bad[:3]:
    a, b = set(int(x) for x in gpu.hits(k)), set(int(x) for x in ora.hits(k))
    print("  guide", k, "gpu", len(a), "oracle", len(b), "missing", len(b - a), "extra", len(a - b), "overflow gpu/oracle", int(gpu.summaries["overflow"][k]), bool(ora.full[k]))
    idx = {int(v): i for i, v in enumerate(t)}
    for v in sorted(b - a)[:5]:
        x = (int(g[k]) ^ v) >> 6
        print("    missing target index", idx[v], "first 3 bases rank", (v >> (6 + 34)) & 63, "xor", hex(x & ((1 << 40) - 1)))
