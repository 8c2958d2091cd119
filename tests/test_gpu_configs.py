"""BASELINE.json's configurations at their stated sizes, HIP path against the CPU oracle through the C ABI.

  C1  one guide (EMX1) against a chr22-scale database (plumbing case of the reference's Quickstart)
  C2  1 000 random NGG 20-mers against a chr22-scale database (4.5e6 unique targets), <= 4 mismatches: hit lists, positions,
      otCount / OVERFLOW and the CFD / Hsu2013 / closest-hit aggregates of EVERY guide, bit for bit against the oracle
      (BlockManager.scala:212-254 is the loop the oracle restates; the oracle needs ~6 s for it)
  a 60-second slice of tools/stress_parity.py (randomised sizes x mismatch budgets x cut-offs), so that the driver's run sees it

C3 (100 000 guides vs 3.0e8 targets) is tests/test_gpu_fullscale.py: the oracle would need ~17 min for it, so it is checked there
through an independent brute-force scan of sampled guides and size-independent properties, and bench.py verifies its own step."""
import time

import numpy as np
import pytest

from flashfry_amd import synth
from tests.helpers import assert_same_hits, assert_same_scores, make_case

pytestmark = pytest.mark.gpu

T_CHR22 = 4_500_000   # chr22 NGG ~ 4-5e6 unique targets
EMX1 = "GAGTCCGAGCAGAAGAAGAAGGG"  # the Quickstart guide (README.md of the reference), 20-mer + NGG


@pytest.fixture(scope="module")
def capi():
    from flashfry_amd import capi
    assert capi.load_library().ffh_device_count() >= 1, "no GPU visible: the gpu tests must run on the MI355X box"
    return capi


@pytest.fixture(scope="module")
def chr22(oracle):
    odb, t, p, g = make_case(oracle, T_CHR22, 1000, enzyme=3, seed=22)
    return odb, t, p, g


def test_config_c2_1000_guides_vs_chr22_scale_is_bit_identical_to_the_oracle(capi, oracle, chr22):
    odb, t, p, g = chr22
    assert len(t) >= T_CHR22 and len(g) == 1000
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        gpu = ctx.discover(g, 4, 2000, jost=True)
        only = ctx.finalize(2000, summaries_only=True, jost=True)
        assert only.summaries.tobytes() == gpu.summaries.tobytes()
    ora = odb.discover(g, 4, 2000)
    assert_same_hits(gpu, ora)                              # hit lists in database order, positions, otCount, OVERFLOW
    assert_same_scores(oracle, 3, g, gpu, ora, jost=True)   # per-hit CFD and the per-guide aggregates, f64 bit for bit
    assert gpu.n_hits > 1000


def test_config_c2_with_a_tight_cutoff_and_five_mismatches(capi, oracle, chr22):
    odb, t, p, g = chr22
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        for max_mm, max_ot in ((5, 25), (3, 2000)):
            gpu = ctx.discover(g[:300], max_mm, max_ot)
            ora = odb.discover(g[:300], max_mm, max_ot)
            assert_same_hits(gpu, ora)
            assert_same_scores(oracle, 3, g[:300], gpu, ora)
            if max_mm == 5:
                assert gpu.summaries["overflow"].sum() > 0


def test_config_c1_one_guide_vs_chr22_scale(capi, oracle, chr22):
    odb, t, p, _ = chr22
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    v = 0
    for ch in EMX1:
        v = (v << 2) | code[ch]
    g = np.array([v | (1 << 48)], dtype=np.uint64)          # BitEncoding.bitEncodeString: count 1 in bits 63:48
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        gpu = ctx.discover(g, 4, 2000)
    ora = odb.discover(g, 4, 2000)
    assert_same_hits(gpu, ora)
    assert_same_scores(oracle, 3, g, gpu, ora)


@pytest.mark.parametrize("checker", ["inproc", "isolated"])
def test_randomised_parity_slice(capi, oracle, checker, monkeypatch):
    """tools/stress_parity.py for a bounded time: seeds x database sizes x guide counts x 0-6 mismatches x cut-offs x ALL SIX
    packs (every third case walks through the enzymes 1 .. 6 in turn).  Once with the oracle in this process and once with it in a
    process of its own that never loads HIP (tests/oracle_proc.py: nothing the library does to host memory can reach that checker);
    the library's page-locked result blocks carry canaries and poison meanwhile (FFH_POOL_DEBUG, ffh_debug_pool_errors)."""
    from tests.test_gpu_parity import dense_case
    from tests.helpers import make_enzyme_case
    monkeypatch.setenv("FFH_POOL_DEBUG", "1")   # (read once per process: effective when this is the first test to create a result)
    if checker == "isolated":
        from tests import oracle_proc
        oracle = oracle_proc.RemoteOracle()
    rng = np.random.default_rng(20260928 if checker == "inproc" else 20260929)
    t0, n, seen = time.time(), 0, set()
    budget = 60.0 if checker == "inproc" else 30.0
    while time.time() - t0 < budget or len(seen) < 6:
        seed = int(rng.integers(0, 1 << 30))
        max_mm = int(rng.choice([0, 1, 2, 3, 4, 4, 4, 5, 6]))
        max_ot = int(rng.choice([5, 40, 60, 300, 2000]))
        enz = 3
        kind = n % 3
        if kind == 2:
            enz = (n // 3) % 6 + 1
            odb, t, p, g = make_enzyme_case(oracle, enz, int(rng.integers(500, 200000)), int(rng.integers(1, 300)), seed=seed)
        elif kind == 0:
            odb, t, p, g = make_case(oracle, int(rng.integers(100, 400000)), int(rng.integers(1, 600)), enzyme=3, seed=seed)
        else:
            ng = int(rng.integers(10, 500))
            odb, t, p, g = dense_case(oracle, n_random=int(rng.integers(1000, 120000)), n_guides=ng,
                                      n_dense=int(rng.integers(1, min(60, ng))), variants=int(rng.integers(10, 200)), seed=seed)
        with capi.Context(enz) as ctx:
            ctx.load_soa(t, p)
            gpu = ctx.discover(g, max_mm, max_ot, jost=True)
            only = ctx.finalize(max_ot, summaries_only=True, jost=True)
            assert only.summaries.tobytes() == gpu.summaries.tobytes()
        ora = odb.discover(g, max_mm, max_ot)
        assert_same_hits(gpu, ora)
        if checker == "isolated":
            oracle.prefetch(enz, g, ora)
        assert_same_scores(oracle, enz, g, gpu, ora, jost=True)
        seen.add(enz)
        n += 1
    assert capi.load_library().ffh_debug_pool_errors() == 0
    assert n >= 12 and seen == {1, 2, 3, 4, 5, 6}, "the slice should get through a few dozen cases in a minute (%d, enzymes %s)" % (n, sorted(seen))


_RAW_UNBOUNDED = {}   # raw hits of the unbounded runs of the repeat-genome test (the parametrisation runs bounding 0 first)


def test_bounded_scan_on_a_uniform_database_changes_nothing(capi, oracle, chr22):
    """ffh_scan_bounded forced on (ffh_set_bounding 1) at config C2: three slabs, (almost) nobody retires -- every hit must be found
    exactly once across the slab boundaries of the prefix image and the per-slab suffix images"""
    odb, t, p, g = chr22
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        ctx.set_bounding(1)
        gpu = ctx.discover(g, 4, 2000, jost=True)
        tm = ctx.timings()
        lim = ctx.discover(g[:200], 5, 30)
        assert ctx.timings().bounded_slabs >= 3
        more = ctx.finalize(31)                               # a larger limit than the scan was bounded by: rescanned unbounded, not refused
        assert ctx.timings().bounded_slabs == 0
        ctx.scan(g[:200], 5)                                  # plain ffh_scan is never bounded: any limit may follow
        un = ctx.finalize(30)
    assert tm.bounded_slabs >= 3
    ora = odb.discover(g, 4, 2000)
    assert_same_hits(gpu, ora)
    assert_same_scores(oracle, 3, g, gpu, ora, jost=True)
    assert_same_hits(lim, odb.discover(g[:200], 5, 30))
    assert_same_hits(more, odb.discover(g[:200], 5, 31))
    assert lim.summaries.tobytes() == un.summaries.tobytes() and np.array_equal(lim.hit_targets, un.hit_targets)


@pytest.mark.parametrize("bounding", [0, 1])
@pytest.mark.parametrize("max_mm,max_ot", [(4, 2000), (4, 100), (5, 2000)])
def test_repeat_structured_genome_with_guides_sampled_from_it(capi, oracle, max_mm, max_ot, bounding):
    """The heavy-tailed case a real genome is (synth.make_repeat_database: repeat families of thousands of near-copies at 1-14 %
    divergence, low-complexity tracts with counts in the hundreds) with the guides drawn FROM the genome by position, so that many
    guides sit inside a family: hundreds to thousands of raw hits per such guide, most of them beyond the ordered cut-off, buckets
    and candidate lists far larger than a compare batch (the kernel's piecewise path).  Bit for bit against the oracle."""
    db = synth.make_repeat_database(1_500_000, seed=synth.DB_SEED + 7)
    g = synth.as_u64(synth.make_guides_from_database(db, 700, seed=synth.GUIDE_SEED + 7))
    t, p = synth.as_u64(db["targets"]), synth.as_u64(db["positions"])
    odb = oracle.db_from_sorted(3, t, p, contigs=synth.CONTIGS_24)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        ctx.set_bounding(bounding)
        gpu = ctx.discover(g, max_mm, max_ot, jost=True)
        only = ctx.finalize(max_ot, summaries_only=True, jost=True)
        assert only.summaries.tobytes() == gpu.summaries.tobytes()
        tm = ctx.timings()
    ora = odb.discover(g, max_mm, max_ot)
    assert_same_hits(gpu, ora)
    assert_same_scores(oracle, 3, g, gpu, ora, jost=True)
    n_over = int(gpu.summaries["overflow"].sum())
    assert n_over >= 20 and n_over < len(g), n_over                       # many guides reach the cut-off, most do not
    if bounding:   # ffh_scan_bounded: guides that reached the limit in an early slab were not scanned against the later ones
        assert tm.bounded_slabs >= 3 and tm.retired_guides > 0
        _RAW_UNBOUNDED.setdefault((max_mm, max_ot), tm.n_raw_hits)
        assert tm.n_raw_hits <= _RAW_UNBOUNDED[(max_mm, max_ot)]
        return
    _RAW_UNBOUNDED[(max_mm, max_ot)] = tm.n_raw_hits
    assert tm.n_raw_hits > 3 * gpu.n_hits or max_ot == 2000              # far more raw hits than retained ones under a tight cut-off
    assert int((t >> np.uint64(48)).max()) >= 100                        # multi-copy targets (low-complexity tracts) are in play


@pytest.mark.parametrize("max_mm,max_ot", [(4, 2000), (4, 60), (5, 500)])
def test_bounded_scan_stops_a_guide_inside_the_slab_where_it_reaches_the_limit(capi, oracle, monkeypatch, max_mm, max_ot):
    """round 5: after every slab of a bounded scan the records of a guide that reaches maximumOffTargets IN that slab are dropped behind
    the part (1/32 of the slab's index span) in which it reaches it (k_slab_totals, k_slab_subhist, k_slab_threshold, k_slab_filter,
    k_slab_keep).  Every dropped record has records of its guide that reach the limit before it in database order, so the delivered
    lists, positions, scores and aggregates are those of the oracle and of the scan with the filter off (FFH_SLAB_FILTER=0), while
    fewer records reach the ordering."""
    db = synth.make_repeat_database(2_500_000, seed=synth.DB_SEED + 31)
    g = synth.as_u64(synth.make_guides_from_database(db, 1500, seed=synth.GUIDE_SEED + 31))
    t, p = synth.as_u64(db["targets"]), synth.as_u64(db["positions"])
    odb = oracle.db_from_sorted(3, t, p, contigs=synth.CONTIGS_24)
    res, raw = {}, {}
    for filt in ("1", "0"):
        monkeypatch.setenv("FFH_SLAB_FILTER", filt)
        with capi.Context(3) as ctx:
            ctx.load_soa(t, p)
            ctx.set_bounding(1)
            res[filt] = ctx.discover(g, max_mm, max_ot, jost=True)
            tm = ctx.timings()
            assert tm.bounded_slabs >= 3
            raw[filt] = tm.n_raw_hits
            again = ctx.discover(g, max_mm, max_ot, summaries_only=True, jost=True)   # (the buffers of the first call reused)
            assert again.summaries.tobytes() == res[filt].summaries.tobytes()
    assert_same_hits(res["1"], odb.discover(g, max_mm, max_ot))
    assert res["1"].summaries.tobytes() == res["0"].summaries.tobytes()
    assert np.array_equal(res["1"].hit_targets, res["0"].hit_targets) and np.array_equal(res["1"].positions, res["0"].positions)
    assert res["1"].n_hits <= raw["1"] < raw["0"], (res["1"].n_hits, raw)


def test_bounded_scan_with_guide_batches_and_other_enzymes(capi, oracle, monkeypatch):
    """the slabs of a bounded scan run on the packed set of guides still active, batch by batch: any batch size gives the oracle's
    result; a 19-mer 3'-PAM pack is bounded as well, a 5'-PAM pack (Cpf1) silently stays unbounded"""
    db = synth.make_repeat_database(400_000, seed=synth.DB_SEED + 11)
    g = synth.as_u64(synth.make_guides_from_database(db, 300, seed=synth.GUIDE_SEED + 11))
    t, p = synth.as_u64(db["targets"]), synth.as_u64(db["positions"])
    odb = oracle.db_from_sorted(3, t, p, contigs=synth.CONTIGS_24)
    ora = odb.discover(g, 4, 150)
    for batch in ("1000000", "53"):
        monkeypatch.setenv("FFH_MAX_GUIDE_BATCH", batch)
        with capi.Context(3) as ctx:
            ctx.load_soa(t, p)
            ctx.set_bounding(1)
            gpu = ctx.discover(g, 4, 150, jost=True)
            tm = ctx.timings()
        assert tm.bounded_slabs == 6 and tm.retired_guides > 0 and tm.compare_launches >= 6
        assert_same_hits(gpu, ora)
        assert_same_scores(oracle, 3, g, gpu, ora, jost=True)
    monkeypatch.delenv("FFH_MAX_GUIDE_BATCH")
    # spCas9-NGG 19-mers (22 bases, 3' PAM, sequence in bits 43:0): bounded like the 23-mers
    rng = np.random.default_rng(66)
    seq = np.unique((rng.integers(0, 1 << 38, size=160_000, dtype=np.uint64) << np.uint64(6)) | (rng.integers(0, 4, size=160_000, dtype=np.uint64) << np.uint64(4)) | np.uint64(0b1010))
    t2 = seq | (np.uint64(1) << np.uint64(48))
    p2 = rng.integers(0, 1 << 27, size=len(t2), dtype=np.uint64) | (np.uint64(3) << np.uint64(32))
    g2 = t2[rng.integers(0, len(t2), size=150)] ^ (rng.integers(0, 4, size=150, dtype=np.uint64) << np.uint64(2 * 9 + 6))   # one base changed (or not)
    odb2 = oracle.db_from_sorted(6, t2, p2, contigs=synth.CONTIGS_24)
    with capi.Context(6) as ctx:
        ctx.load_soa(t2, p2)
        ctx.set_bounding(1)
        gpu = ctx.discover(g2, 4, 3)
        assert ctx.timings().bounded_slabs == 6
    assert_same_hits(gpu, odb2.discover(g2, 4, 3))
    # neither of these can be bounded, both silently stay plain scans: make_case's 22-base packs carry bits above the sequence, which
    # lead their order (checked by k_slab_cuts); Cpf1 has a 5' PAM, its database order is (bin = the 7 bases after the PAM, sequence)
    odb3, t3, p3, g3 = make_case(oracle, 150_000, 120, enzyme=5, seed=5)
    from tests.test_gpu_parity import _cas12a_database
    g40 = rng.integers(0, 1 << 40, size=120, dtype=np.uint64)
    t4, p4 = _cas12a_database(rng, 150_000, g40, 6)
    g4 = g40 | np.uint64(0b11111100 << 40) | np.uint64(1 << 48)
    odb4 = oracle.db_from_sorted(1, t4, p4, contigs=synth.CONTIGS_24)
    for enzyme, odb_, t_, p_, g_, max_ot in ((5, odb3, t3, p3, g3, 2000), (1, odb4, t4, p4, g4, 3)):
        with capi.Context(enzyme) as ctx:
            ctx.load_soa(t_, p_)
            ctx.set_bounding(1)
            gpu = ctx.discover(g_, 4, max_ot)
            assert ctx.timings().bounded_slabs == 0
        ora_ = odb_.discover(g_, 4, max_ot)
        assert_same_hits(gpu, ora_)
        assert gpu.n_hits > 0


@pytest.mark.parametrize("sort", ["seg", "lsd", "bin", "lsd-onesweep"])
def test_hit_ordering_by_segments_and_by_six_passes_agree_with_the_oracle(capi, oracle, monkeypatch, sort):
    """the hits are ordered by two device-wide passes over the guide bits + one wave per guide ranking its segment (k_segsort), guides
    with more than 1024 raw hits by a block-level sort of their own (k_segsort_heavy); scans with many hits per guide take the
    six-pass LSD sort; a moderate number of hits takes one most-significant-digit pass + one in-LDS launch per bin of guides (round 5:
    k_msd_scatter, k_binsort; bins that outgrow LDS go through k_binsort_heavy).  FFH_SORT forces each: all must give the oracle's lists on a genome whose repeat families put thousands of
    raw hits on some guides and a handful on others."""
    monkeypatch.setenv("FFH_SORT", sort.split("-")[0])
    if sort.endswith("onesweep"):
        monkeypatch.setenv("FFH_ONESWEEP", "1")       # the LSD passes with decoupled look-back (off by default: profiles/r05/ab_log.txt 8)
    db = synth.make_repeat_database(900_000, seed=synth.DB_SEED + 21)
    g = synth.as_u64(synth.make_guides_from_database(db, 500, seed=synth.GUIDE_SEED + 21))
    t, p = synth.as_u64(db["targets"]), synth.as_u64(db["positions"])
    odb = oracle.db_from_sorted(3, t, p, contigs=synth.CONTIGS_24)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        ctx.set_bounding(0)
        gpu = ctx.discover(g, 4, 2 ** 31 - 1, jost=True)                 # no cut-off: every raw hit is delivered, in database order
        tm = ctx.timings()
        cut = ctx.discover(g, 5, 300)
    per_guide = np.diff(gpu.guide_offsets.astype(np.int64))
    assert per_guide.max() > 1024 and (per_guide <= 128).sum() > 50 and ((per_guide > 128) & (per_guide <= 1024)).sum() > 5, (per_guide.max(), tm.n_raw_hits)
    assert_same_hits(gpu, odb.discover(g, 4, 2 ** 31 - 1))
    assert_same_hits(cut, odb.discover(g, 5, 300))


@pytest.mark.parametrize("n_guides,variants", [(1, 1500), (9, 1100), (9, 3000), (40, 1200), (300, 600)])
def test_scans_made_of_a_few_large_guides(capi, oracle, n_guides, variants):
    """a handful of guides with a thousand hits each (a five-mismatch look-up of ten guides): one block of k_binsort<true> holds the
    whole scan (<= 12 288 records) or a bin of it, and its 16 waves share every guide -- chunks of 256 indices ordered in registers,
    then ranked against the guide's other chunks by bisection.  9 x 1 100 fits the single block, 9 x 3 000 and 40 x 1 200 take the
    digit pass first (a few guides per bin), 300 x 600 mixes guides above and below the 256-hit network.  No cut-off: every hit is
    delivered, so the order of every segment is checked."""
    from tests.test_gpu_parity import dense_case
    odb, t, p, g = dense_case(oracle, n_random=40_000, n_guides=n_guides, n_dense=n_guides, variants=variants, seed=70 + n_guides)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        ctx.set_bounding(0)
        gpu = ctx.discover(g, 4, 2 ** 31 - 1, jost=True)
        cut = ctx.discover(g, 3, 50)
    per_guide = np.diff(gpu.guide_offsets.astype(np.int64))
    assert per_guide.max() > 256 and per_guide.mean() > 256, per_guide
    assert_same_hits(gpu, odb.discover(g, 4, 2 ** 31 - 1))
    assert_same_hits(cut, odb.discover(g, 3, 50))


def test_two_part_pipelined_discover_delivers_the_same_result(capi, oracle, monkeypatch):
    """FFH_PIPELINE=1 (off by default: slower on this stack, profiles/r05/ab_log.txt 7): a list-delivering ffh_discover scans 60 % of
    the guides, leaves their lists on the copy stream, scans the rest and delivers both parts in ONE result block.  Every array must be
    what the unsplit call delivers, with and without positions / per-hit scores, cut-off far and biting."""
    from tests.test_gpu_parity import dense_case
    odb, t, p, g = dense_case(oracle, n_random=150_000, n_guides=600, n_dense=50, variants=150, seed=61)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        plain = {(mo, kw): ctx.discover(g, 4, mo, jost=True, **dict(kw)) for mo in (2000, 30) for kw in ((), (("positions", False), ("hit_scores", False)))}
    monkeypatch.setenv("FFH_PIPELINE", "1")
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        for (mo, kw), want in plain.items():
            got = ctx.discover(g, 4, mo, jost=True, **dict(kw))
            assert got.n_hits == want.n_hits and got.n_positions == want.n_positions
            assert got.summaries.tobytes() == want.summaries.tobytes()
            for name in ("guide_offsets", "hit_targets", "hit_mismatches"):
                assert np.array_equal(getattr(got, name), getattr(want, name)), name
            if not kw:
                assert np.array_equal(got.positions, want.positions) and np.array_equal(got.pos_offsets, want.pos_offsets)
                assert np.array_equal(np.isnan(got.hit_cfd), np.isnan(want.hit_cfd)) and np.array_equal(np.nan_to_num(got.hit_cfd), np.nan_to_num(want.hit_cfd))
        assert_same_hits(ctx.discover(g, 4, 2000), odb.discover(g, 4, 2000))
    monkeypatch.delenv("FFH_PIPELINE")
