#!/usr/bin/env python3
"""Randomised parity sweep against the oracle (dev tool, GPU box): many seeds x sizes x mismatch budgets x cut-offs through the
same helpers as tests/test_gpu_parity.py.  Prints one line per case and stops at the first difference.

  python tools/stress_parity.py [seconds [seed [first case]]] [--oracle inproc|isolated] [--quiet]

--oracle isolated (round 4): the oracle runs in a PROCESS OF ITS OWN (tests/oracle_proc.py) that never loads HIP -- no device write,
page-locked block or host thread of the library can reach the checker's memory.  --oracle inproc keeps the checker in this process,
next to the library (the form every earlier sweep had).  Either way FFH_POOL_DEBUG=1 is set: the library's page-locked result blocks
carry canaries, released blocks a poison pattern (PinnedPool, ffh_api.hip); ffh_debug_pool_errors() must stay 0."""
import os, sys, time
os.environ.setdefault("FFH_POOL_DEBUG", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flashfry_amd import capi
import oracle_lib
from helpers import make_case, make_enzyme_case, assert_same_hits, assert_same_scores
import stress_cases

args = [a for a in sys.argv[1:] if not a.startswith("--")]
mode = "inproc"
if "--oracle" in sys.argv:
    mode = sys.argv[sys.argv.index("--oracle") + 1]
    args.remove(mode)
quiet = "--quiet" in sys.argv
seal = "--seal" in sys.argv
assert mode in ("inproc", "isolated")
if mode == "isolated":
    import oracle_proc
    oracle = oracle_proc.RemoteOracle()
else:
    oracle = oracle_lib.load()
L = capi.load_library()
# tools/heapwatch.c preloaded (LD_PRELOAD): every freed chunk of the process is parked, poisoned and verified -- after every case here
import ctypes
try:
    _hw = ctypes.CDLL(None)
    _hw.heapwatch_check_all.restype = ctypes.c_ulonglong
    heapwatch = _hw.heapwatch_check_all
except AttributeError:
    heapwatch = None
hw_seen = 0
budget = float(args[0]) if len(args) > 0 else 120.0
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 12345)
t0, n = time.time(), 0
start = int(args[2]) if len(args) > 2 else 1      # replay: skip the cases before this one (same random draws, nothing built)
by_kind = {}
while time.time() - t0 < budget:
    c = stress_cases.draw(rng)   # (the draws, in their order: tools/stress_cases.py -- tools/oracle_case_dump.py names the same cases)
    seed, kind, enz, max_mm, max_ot, par, bounding, pos, sc = (c[k] for k in ("seed", "kind", "enz", "max_mm", "max_ot", "par", "bounding", "pos", "sc"))
    if n + 1 < start:
        n += 1
        continue
    if not quiet:
        print("case %d seed %d kind %d enzyme %d mm %d max_ot %d par %s bounding %d" % (n + 1, seed, kind, enz, max_mm, max_ot, par, bounding), flush=True)
    odb, t, p, g = stress_cases.build(oracle, c)
    if seal and hasattr(odb, "seal"):
        odb.seal()   # the checker's database in ONE read-only mapping: a stray CPU store into it faults on the spot (heapwatch prints the backtrace)
    # the inputs as the checker and the library were given them: they must not change under either
    t_sum, p_sum, g_copy = int(t.sum(dtype=np.uint64)), int(p.sum(dtype=np.uint64)), g.copy()
    # ... and neither must the checker's own database (in-process checker only: round 5 saw ONE case in 15 810 where the in-process
    # database answered with every hit of a bin twice and differently when asked again, while a fresh database and the isolated checker
    # agreed with the library -- a write into the checker's heap, or the checker reading memory it never wrote?  The checksums say which)
    watch = mode == "inproc" and hasattr(odb, "checksums")
    db_sum, db_per = odb.checksums() if watch else (0, None)

    def db_state(when):
        if not watch:
            return True
        now, _ = odb.checksums()
        if now == db_sum:
            return True
        print("ORACLE DATABASE CHANGED %s: case %d, first changed bin %d of %d" % (when, n + 1, odb.first_changed_bin(db_per), odb.n_bins), flush=True)
        return False
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
    assert db_state("while the library ran"), "the checker's in-process database changed while the library ran"
    ora = odb.discover(g, max_mm, max_ot)
    db_same = db_state("during the checker's own discover")
    if not db_same or not np.array_equal(gpu.guide_offsets, ora.guide_offsets) or not np.array_equal(gpu.hit_targets, ora.hit_targets):
        # what differs, and whether the same context gives the same answer when asked again (a stale buffer or a race would not)
        bad = [k for k in range(len(g)) if not np.array_equal(gpu.hits(k), ora.hits(k))]
        print("MISMATCH case %d (%s oracle): %d guides differ: %s" % (n + 1, mode, len(bad), bad[:10]), flush=True)
        print("  inputs unchanged: targets %s positions %s guides %s; %d distinct guides of %d" % (
            int(t.sum(dtype=np.uint64)) == t_sum, int(p.sum(dtype=np.uint64)) == p_sum, np.array_equal(g, g_copy), len(np.unique(g)), len(g)), flush=True)
        idx = {int(v): i for i, v in enumerate(t)}
        for k in bad[:6]:
            a, b = [int(x) for x in gpu.hits(k)], [int(x) for x in ora.hits(k)]
            print("  guide %d (%016x): gpu list %s, oracle list %s (database indices), overflow gpu %d oracle %d, gpu ot_count %d" % (
                k, int(g[k]), [idx[v] for v in a][:8], [idx.get(v, -1) for v in b][:8], int(gpu.summaries["overflow"][k]), int(ora.full[k]), int(gpu.summaries["ot_count"][k])), flush=True)
        ora_again = odb.discover(g, max_mm, max_ot)
        print("  the same oracle database asked again: %s%s; its memory is %s" % (
            "same answer" if np.array_equal(ora_again.guide_offsets, ora.guide_offsets) and np.array_equal(ora_again.hit_targets, ora.hit_targets) else "ANOTHER answer",
            " (the library's)" if np.array_equal(ora_again.guide_offsets, gpu.guide_offsets) and np.array_equal(ora_again.hit_targets, gpu.hit_targets) else "",
            "as it was built" if db_state("by the time of the second question") else "NOT as it was built"), flush=True)
        for tag, orc in (("a fresh in-process oracle database", oracle_lib.load()), ("a fresh isolated oracle", __import__("oracle_proc").RemoteOracle())):
            o2 = orc.db_from_sorted(enz, t, p, contigs=["c1"]).discover(g, max_mm, max_ot)
            print("  %s: %s" % (tag, "agrees with the library" if np.array_equal(o2.guide_offsets, gpu.guide_offsets) and np.array_equal(o2.hit_targets, gpu.hit_targets) else
                                "agrees with the first oracle answer" if np.array_equal(o2.guide_offsets, ora.guide_offsets) and np.array_equal(o2.hit_targets, ora.hit_targets) else "a third answer"), flush=True)
        for tag, bnd in (("again, same context settings", bounding), ("unbounded", 0)):
            with capi.Context(enz) as ctx2:
                ctx2.load_soa(t, p)
                ctx2.set_bounding(bnd)
                r2 = ctx2.discover(g, max_mm, max_ot, jost=True)
                print("  library %s: %s" % (tag, "agrees with the oracle" if np.array_equal(r2.guide_offsets, ora.guide_offsets) and np.array_equal(r2.hit_targets, ora.hit_targets) else
                                            "differs (%d guides), %s its first answer" % (sum(1 for k in range(len(g)) if not np.array_equal(r2.hits(k), ora.hits(k))),
                                                                                        "equal to" if np.array_equal(r2.hit_targets, gpu.hit_targets) else "NOT equal to")), flush=True)
        print("  pool errors so far: %d" % L.ffh_debug_pool_errors(), flush=True)
    assert_same_hits(gpu, ora)
    assert db_same, "the checker's in-process database changed"
    if mode == "isolated":
        oracle.prefetch(enz, g, ora)
    assert_same_scores(oracle, enz, g, gpu, ora, jost=True)
    assert int(t.sum(dtype=np.uint64)) == t_sum and int(p.sum(dtype=np.uint64)) == p_sum and np.array_equal(g, g_copy), "an input array changed during the case"
    assert L.ffh_debug_pool_errors() == 0, "the page-locked pool checks fired"
    if heapwatch is not None:
        e = int(heapwatch())
        if e != hw_seen:
            print("HEAPWATCH: %d damaged chunk(s) found by the end of case %d (kind %d enzyme %d mm %d max_ot %d par %s bounding %d)" % (e - hw_seen, n + 1, kind, enz, max_mm, max_ot, par, bounding), flush=True)
            hw_seen = e
    n += 1
    by_kind[(kind, enz)] = by_kind.get((kind, enz), 0) + 1
    if not quiet or n % 500 == 0:
        print("ok %3d kind %d enzyme %d T %7d G %4d mm %d max_ot %4d hits %8d bounding %2d slabs %d overflow %d" % (n, kind, enz, len(t), len(g), max_mm, max_ot, gpu.n_hits, bounding, slabs,
                                                                                                                      int(gpu.summaries["overflow"].sum())), flush=True)
del gpu, only, lean
import gc
gc.collect()
print("all %d cases agree (%s oracle, pool errors %d, heapwatch %s, %.0f s)" % (n - (start - 1), mode, L.ffh_debug_pool_errors(), "off" if heapwatch is None else "%d damaged chunks" % hw_seen, time.time() - t0))
print("cases by (kind, enzyme):", dict(sorted(by_kind.items())))
