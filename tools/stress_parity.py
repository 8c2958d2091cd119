#!/usr/bin/env python3
"""Randomised parity sweep against the oracle (dev tool, GPU box): many seeds x sizes x mismatch budgets x cut-offs through the
same helpers as tests/test_gpu_parity.py.  Prints one line per case and stops at the first difference.
  python tools/stress_parity.py [seconds [seed]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flashfry_amd import capi
import oracle_lib
from helpers import make_case, make_enzyme_case, assert_same_hits, assert_same_scores
from test_gpu_parity import dense_case

oracle = oracle_lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t0, n = time.time(), 0
start = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # replay: skip the cases before this one (same random draws, nothing built)
while time.time() - t0 < budget:
    seed = int(rng.integers(0, 1 << 30))
    kind = int(rng.integers(0, 4))
    enz = 3
    max_mm = int(rng.choice([0, 1, 2, 3, 4, 4, 4, 5, 6]))
    max_ot = int(rng.choice([5, 40, 60, 300, 2000]))
    if kind == 0:
        par = (int(rng.integers(100, 400000)), int(rng.integers(1, 600)))
    elif kind == 2:   # repeat-structured genome, guides sampled from it (families, multi-copy targets, many OVERFLOW guides)
        par = (int(rng.integers(70000, 900000)), float(rng.uniform(0.1, 0.6)), int(rng.integers(20, 400)))
        max_mm = min(max_mm, 5)
    elif kind == 3:   # any of the six packs (Cpf1's 5' PAM and bin order, NAG, the 19-mers with their 7 .. 12-base rest keys)
        enz = int(rng.integers(1, 7))
        par = (int(rng.integers(500, 300000)), int(rng.integers(1, 400)))
    else:
        ng = int(rng.integers(10, 500))
        par = (int(rng.integers(1000, 120000)), ng, int(rng.integers(1, min(60, ng))), int(rng.integers(10, 200)))
    bounding = int(rng.choice([-1, 0, 1, 1]))     # ffh_scan_bounded engages for databases of >= 65536 targets
    pos, sc = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    if n + 1 < start:
        n += 1
        continue
    print("case %d seed %d kind %d enzyme %d mm %d max_ot %d par %s bounding %d" % (n + 1, seed, kind, enz, max_mm, max_ot, par, bounding), flush=True)
    if kind == 0:
        odb, t, p, g = make_case(oracle, par[0], par[1], enzyme=3, seed=seed)
    elif kind == 2:
        from flashfry_amd import synth
        db = synth.make_repeat_database(par[0], seed=seed, repeat_fraction=par[1])
        g = synth.as_u64(synth.make_guides_from_database(db, par[2], seed=seed + 1))
        t, p = synth.as_u64(db["targets"]), synth.as_u64(db["positions"])
        odb = oracle.db_from_sorted(3, t, p, contigs=synth.CONTIGS_24)
    elif kind == 3:
        odb, t, p, g = make_enzyme_case(oracle, enz, par[0], par[1], seed=seed)
    else:
        odb, t, p, g = dense_case(oracle, n_random=par[0], n_guides=par[1], n_dense=par[2], variants=par[3], seed=seed)
    with capi.Context(enz) as ctx:
        ctx.load_soa(t, p)
        ctx.set_bounding(bounding)
        gpu = ctx.discover(g, max_mm, max_ot, jost=True)
        slabs = ctx.timings().bounded_slabs
        only = ctx.finalize(max_ot, summaries_only=True, jost=True)
        assert only.summaries.tobytes() == gpu.summaries.tobytes()
        lean = ctx.finalize(max_ot, jost=True, positions=pos, hit_scores=sc)   # the list path with its optional arrays left out
        assert lean.summaries.tobytes() == gpu.summaries.tobytes() and np.array_equal(lean.hit_targets, gpu.hit_targets)
        assert np.array_equal(lean.hit_mismatches, gpu.hit_mismatches) and np.array_equal(lean.guide_offsets, gpu.guide_offsets)
        if pos:
            assert np.array_equal(lean.positions, gpu.positions) and np.array_equal(lean.pos_offsets, gpu.pos_offsets)
        if sc:
            assert lean.hit_cfd.tobytes() == gpu.hit_cfd.tobytes()
    ora = odb.discover(g, max_mm, max_ot)
    if not np.array_equal(gpu.guide_offsets, ora.guide_offsets) or not np.array_equal(gpu.hit_targets, ora.hit_targets):
        # what differs, and whether the same context gives the same answer when asked again (a stale buffer or a race would not)
        bad = [k for k in range(len(g)) if not np.array_equal(gpu.hits(k), ora.hits(k))]
        print("MISMATCH case %d: %d guides differ: %s" % (n + 1, len(bad), bad[:10]), flush=True)
        idx = {int(v): i for i, v in enumerate(t)}
        for k in bad[:4]:
            a, b = set(int(x) for x in gpu.hits(k)), set(int(x) for x in ora.hits(k))
            print("  guide %d: gpu %d hits, oracle %d, missing %s extra %s (database indices), overflow gpu %d oracle %d, gpu ot_count %d" % (
                k, len(a), len(b), sorted(idx[v] for v in b - a)[:8], sorted(idx[v] for v in a - b)[:8], int(gpu.summaries["overflow"][k]), int(ora.full[k]), int(gpu.summaries["ot_count"][k])), flush=True)
        ora_again = odb.discover(g, max_mm, max_ot)
        print("  the oracle asked again: %s; the guide array holds %d distinct guides of %d" % (
            "same answer" if np.array_equal(ora_again.guide_offsets, ora.guide_offsets) and np.array_equal(ora_again.hit_targets, ora.hit_targets) else "ANOTHER answer",
            len(np.unique(g)), len(g)), flush=True)
        for tag, bnd in (("again, same context settings", bounding), ("unbounded", 0)):
            with capi.Context(enz) as ctx2:
                ctx2.load_soa(t, p)
                ctx2.set_bounding(bnd)
                r2 = ctx2.discover(g, max_mm, max_ot, jost=True)
                print("  %s: %s" % (tag, "agrees with the oracle" if np.array_equal(r2.guide_offsets, ora.guide_offsets) and np.array_equal(r2.hit_targets, ora.hit_targets) else
                                    "differs (%d guides)" % sum(1 for k in range(len(g)) if not np.array_equal(r2.hits(k), ora.hits(k)))), flush=True)
    assert_same_hits(gpu, ora)
    assert_same_scores(oracle, enz, g, gpu, ora, jost=True)
    n += 1
    print("ok %3d kind %d enzyme %d T %7d G %4d mm %d max_ot %4d hits %8d bounding %2d slabs %d overflow %d" % (n, kind, enz, len(t), len(g), max_mm, max_ot, gpu.n_hits, bounding, slabs,
                                                                                                                  int(gpu.summaries["overflow"].sum())), flush=True)
print("all %d cases agree" % n)
