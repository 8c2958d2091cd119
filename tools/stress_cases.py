"""The case stream of tools/stress_parity.py as a generator (round 6): the same random draws in the same order, so that a case number of
a sweep log names the same inputs here.  Used by tools/oracle_case_dump.py (inputs of a case range -> files for the checker's sanitizer
replays, tools/oracle_replay.c) and by stress_parity.py itself."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def draw(rng):
    """one case's parameters: exactly the draws of tools/stress_parity.py, in its order"""
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
    return dict(seed=seed, kind=kind, enz=enz, max_mm=max_mm, max_ot=max_ot, par=par, bounding=bounding, pos=pos, sc=sc)


def build(oracle, c):
    """(oracle database, targets, positions, guides) of a drawn case"""
    from helpers import make_case, make_enzyme_case
    from flashfry_amd import synth
    kind, par, seed = c["kind"], c["par"], c["seed"]
    if kind == 0:
        return make_case(oracle, par[0], par[1], enzyme=3, seed=seed)
    if kind == 2:
        db = synth.make_repeat_database(par[0], seed=seed, repeat_fraction=par[1])
        g = synth.as_u64(synth.make_guides_from_database(db, par[2], seed=seed + 1))
        t, p = synth.as_u64(db["targets"]), synth.as_u64(db["positions"])
        return oracle.db_from_sorted(3, t, p, contigs=synth.CONTIGS_24), t, p, g
    if kind == 3:
        return make_enzyme_case(oracle, c["enz"], par[0], par[1], seed=seed)
    from test_gpu_parity import dense_case
    return dense_case(oracle, n_random=par[0], n_guides=par[1], n_dense=par[2], variants=par[3], seed=seed)
