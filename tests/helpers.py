"""Shared helpers of the parity tests (oracle vs HIP path)."""
import numpy as np

from flashfry_amd import synth


def make_case(oracle, n_targets, n_guides, enzyme=3, seed=0, max_linear=500, plant=True):
    """synthetic database + guides; returns (oracle db, targets u64, positions u64, guides u64)"""
    g = synth.make_guides(n_guides, seed=synth.GUIDE_SEED + seed)
    db = synth.make_database(n_targets, seed=synth.DB_SEED + seed, plant_guides=g if plant else None)
    targets, positions = synth.as_u64(db["targets"]), synth.as_u64(db["positions"])
    odb = oracle.db_from_sorted(enzyme, targets, positions, bin_width=7, max_linear=max_linear, contigs=synth.CONTIGS_24)
    return odb, targets, positions, synth.as_u64(g)


SCAN_LEN = {1: 24, 2: 23, 3: 23, 4: 23, 5: 22, 6: 22}   # StandardScanParameters.scala:90-215


def make_enzyme_case(oracle, enzyme, n_targets, n_guides, seed=0, max_mut=5):
    """a database of random sites of ANY of the six packs in the reference's database order (3' PAM: by sequence; Cpf1: bin = the 7
    bases after the 5' PAM, then sequence) with counts 1..3, and guides that are mutated database members (0 .. max_mut - 1
    substitutions anywhere in the site), so that hits exist at every level.  Returns (oracle db, targets, positions, guides)."""
    rng = np.random.default_rng(1000 * enzyme + seed)
    L = SCAN_LEN[enzyme]
    raw = np.unique(rng.integers(0, 1 << (2 * L), size=n_targets, dtype=np.uint64))
    counts = rng.integers(1, 4, size=len(raw)).astype(np.uint64)
    targets = raw | (counts << np.uint64(48))
    if enzyme == 1:
        binkey = (raw >> np.uint64(2 * (24 - 11))) & np.uint64(0x3FFF)
        order = np.lexsort((raw, binkey))
        targets, raw = targets[order], raw[order]
    n_pos = int((targets >> np.uint64(48)).sum())
    positions = rng.integers(0, 1 << 27, size=n_pos, dtype=np.uint64) | (np.uint64(L) << np.uint64(52)) | (np.uint64(1) << np.uint64(32))
    odb = oracle.db_from_sorted(enzyme, targets, positions, contigs=["c1"])
    guides = raw[rng.integers(0, len(raw), size=n_guides)].copy()
    for k in range(len(guides)):
        for _ in range(int(rng.integers(0, max_mut))):
            guides[k] ^= np.uint64(int(rng.integers(1, 4)) << (2 * int(rng.integers(0, L))))
    guides |= np.uint64(1) << np.uint64(48)
    return odb, targets, positions, guides


def assert_same_hits(gpu, ora, targets=None):
    """bit-identical hit sets: same retained targets per guide in the same (database) order, same positions,
    same totals and overflow flags"""
    assert gpu.n_guides == ora.n_guides
    assert np.array_equal(gpu.guide_offsets, ora.guide_offsets), "per-guide hit counts differ"
    assert np.array_equal(gpu.hit_targets, ora.hit_targets), "hit targets differ"
    assert np.array_equal(gpu.pos_offsets, ora.pos_offsets), "position offsets differ"
    assert np.array_equal(gpu.positions, ora.positions), "positions differ"
    assert np.array_equal(gpu.summaries["ot_count"].astype(np.int64), ora.current_total), "otCount differs"
    assert np.array_equal(gpu.summaries["overflow"].astype(bool), ora.full), "OVERFLOW flags differ"
    assert np.array_equal(gpu.summaries["n_hits"].astype(np.uint64), np.diff(ora.guide_offsets))


def assert_same_scores(oracle, enzyme, guides, gpu, ora, exact=True, jost=False):
    """per-guide CFD / Hsu2013 / closest-hit / in-genome aggregates against the oracle's string-level restatement"""
    for g in range(ora.n_guides):
        s, per = oracle.score_guide(enzyme, int(guides[g]), ora.hits(g))
        m = gpu.summaries[g]
        assert list(m["hist"]) == list(s.hist)
        assert int(m["closest"]) == (0xFFFFFFFF if s.closest == 2 ** 31 - 1 else s.closest)
        assert int(m["closest_count"]) == s.closest_count
        assert int(m["in_genome"]) == s.in_genome
        if jost:  # JostAndSantosCRISPRi aggregates (requested with FFH_FINALIZE_JOST), bit-identical f64
            if s.jost_valid:
                assert float(m["jost_max"]) == s.jost_max and 1.0 / (1.0 + float(m["jost_sum"])) == s.jost_spec, (g, m, s.jost_max, s.jost_spec)
            else:
                assert float(m["jost_max"]) == 0.0 and float(m["jost_sum"]) == 0.0
        if s.cfd_valid:
            assert gpu.scores_valid
            a, b = int(gpu.guide_offsets[g]), int(gpu.guide_offsets[g + 1])
            got = gpu.hit_cfd[a:b]
            assert np.array_equal(np.isnan(got), np.isnan(per))
            spec = 1.0 / (1.0 + float(m["cfd_sum"]))
            hsu = (100.0 / (100.0 + float(m["hsu_sum"]))) * 100.0
            if exact:
                assert np.array_equal(got[~np.isnan(got)], per[~np.isnan(per)]), "per-hit CFD differs"
                assert float(m["cfd_max"]) == s.cfd_max and spec == s.cfd_spec and hsu == s.hsu, (g, m, s.cfd_max, s.cfd_spec, s.hsu)
            else:  # north-star tolerance: within 1e-6
                assert np.allclose(got[~np.isnan(got)], per[~np.isnan(per)], rtol=0, atol=1e-6)
                assert abs(float(m["cfd_max"]) - s.cfd_max) <= 1e-6 and abs(spec - s.cfd_spec) <= 1e-6 and abs(hsu - s.hsu) <= 1e-6
        else:
            assert not gpu.scores_valid
