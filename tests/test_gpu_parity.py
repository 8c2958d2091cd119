"""Parity of the HIP path against the CPU oracle, through the C ABI (ctypes).  Needs a real MI355X."""
import os

import numpy as np
import pytest

from flashfry_amd import synth
from tests.helpers import assert_same_hits, assert_same_scores, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from flashfry_amd import capi
    capi.load_library()
    assert capi.load_library().ffh_device_count() >= 1, "no GPU visible: the gpu tests must run on the MI355X box"
    return capi


def run_both(capi, oracle, odb, targets, positions, guides, enzyme, max_mm, max_ot, via="soa", plan=None):
    with capi.Context(enzyme) as ctx:
        if plan:
            ctx.set_plan(*plan[:1])
        if via == "soa":
            ctx.load_soa(targets, positions)
        else:
            longs, offs = odb.all_blocks()
            ctx.load_blocks(longs, offs)
        if plan:
            ctx.set_plan(*plan)
        gpu = ctx.discover(guides, max_mm, max_ot, jost=True)
        tm = ctx.timings()
        # the fused aggregates-only epilogue (what bench.py times) must agree with the list-delivering path bit for bit
        only = ctx.finalize(max_ot, summaries_only=True, jost=True)
        assert only.summaries.tobytes() == gpu.summaries.tobytes() and np.array_equal(only.guide_offsets, gpu.guide_offsets)
        assert only.n_hits == gpu.n_hits
    ora = odb.discover(guides, max_mm, max_ot)
    return gpu, ora, tm


@pytest.mark.parametrize("max_mm", [0, 1, 2, 3, 4, 5])
def test_hits_match_oracle_cas9(capi, oracle, max_mm):
    odb, t, p, g = make_case(oracle, 60000, 400, enzyme=3, seed=max_mm)
    gpu, ora, tm = run_both(capi, oracle, odb, t, p, g, 3, max_mm, 2000)
    assert_same_hits(gpu, ora)
    assert_same_scores(oracle, 3, g, gpu, ora, jost=True)
    assert gpu.n_hits > 0


def test_blocks_loader_equals_soa_loader(capi, oracle):
    # max_linear=18 (about the mean bin size) forces a mix of linear and indexed blocks (BlockManager.scala:63-90 dispatch)
    odb, t, p, g = make_case(oracle, 300000, 300, enzyme=2, seed=11, max_linear=18)
    kinds = {int(odb.bin(b)[0][0]) for b in range(0, odb.n_bins, 97)}
    assert kinds == {1, 2}
    a, ora, _ = run_both(capi, oracle, odb, t, p, g, 2, 4, 2000, via="blocks")
    b, _, _ = run_both(capi, oracle, odb, t, p, g, 2, 4, 2000, via="soa")
    assert_same_hits(a, ora)
    assert_same_hits(b, ora)
    assert_same_scores(oracle, 2, g, a, ora, jost=True)


def test_device_block_decoder_refuses_malformed_payloads(capi, oracle):
    """the bin payloads are decoded on the device (ffh_ingest.hpp); the checks of BlockManager.scala:72,85-87,160-170,232-236 hold there"""
    odb, t, p, g = make_case(oracle, 20000, 50, enzyme=3, seed=5, max_linear=2)
    longs, offs = odb.all_blocks()
    with capi.Context(3) as ctx:
        ctx.load_blocks(longs, offs)
        info = ctx.info()
        assert (info.n_targets, info.n_positions) == (len(t), len(p))
        st = ctx.load_stats()
        assert st.raw_bytes == 8 * len(longs) and st.decode_ms > 0
    tgt = np.int64((int(oracle.encode("ACGTACGTACGTACGTACGTAGG")) & ((1 << 48) - 1)) | (2 << 48))
    lin = np.array([1, tgt, 11, 12], dtype=np.int64)                         # one target with two positions
    table = np.full(256, np.int64(-1) << np.int64(32), dtype=np.int64)       # pos = -1, size = 0 everywhere ...
    table[27] = (0 << 32) | 3                                                # ... but sub-bin 27 holds the 3 payload longs
    idx = np.concatenate([[2], table, lin[1:]]).astype(np.int64)

    def load(blocks):
        offs = np.zeros(len(blocks) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(b) for b in blocks])
        with capi.Context(3) as ctx:
            ctx.load_blocks(np.concatenate(blocks).astype(np.int64) if offs[-1] else np.zeros(1, np.int64), offs)
            return ctx.info()

    assert load([lin]).n_targets == 1 and load([lin]).n_positions == 2
    assert load([idx, lin]).n_targets == 2 and load([idx, lin]).n_positions == 4
    assert load([np.array([1], dtype=np.int64)]).n_targets == 0              # a bin without targets is a bare type long
    cases = [
        ([np.array([7, 1, 2], dtype=np.int64)], "Invalid bin type, unknown value: 7"),
        ([np.zeros(0, dtype=np.int64)], "empty block for bin 0"),
        ([lin, np.zeros(0, dtype=np.int64)], "empty block for bin 1"),
        ([np.array([1, tgt & np.int64((1 << 48) - 1), 11], dtype=np.int64)], "Encoded position count should be greater than zero"),
        ([lin[:3]], "exceeds the buffer size"),
        ([idx[:100]], "shorter than its lookup table"),
        ([idx[:-1]], "sub-bin slice out of range"),
        ([np.concatenate([idx, [5]]).astype(np.int64)], "do not cover the payload"),
        ([lin, lin[:3], np.array([9], dtype=np.int64)], "exceeds the buffer size"),    # the first bad bin in database order is reported
        ([lin, np.array([9], dtype=np.int64), lin[:3]], "unknown value: 9"),
    ]
    two = table.copy()
    two[27] = (0 << 32) | 2                                                  # the slice cuts the record in two
    two[28] = (2 << 32) | 1
    cases.append(([np.concatenate([[2], two, lin[1:]]).astype(np.int64)], "exceeds the buffer size"))
    gap = table.copy()
    gap[27] = (0 << 32) | 1
    gap[28] = (1 << 32) | 1
    gap[29] = (3 << 32) | 1                                                  # 1 + 1 != 3: compareIndexedBlock's contiguity assert (:167-168)
    cases.append(([np.concatenate([[2], gap, lin[1:]]).astype(np.int64)], "not contiguous"))
    for blocks, msg in cases:
        with pytest.raises(capi.FlashFryHipError, match=msg):
            load(blocks)


@pytest.mark.parametrize("plan", [(8, 0), (8, 1), (8, 2), (8, 3), (8, 4), (10, 2), (11, 1), (12, 2), (9, 3)])
def test_every_candidate_split_gives_the_same_hits(capi, oracle, plan):
    """the prefix/suffix ball split is an exact filter: any (width, radius) must reproduce the oracle's hit set"""
    odb, t, p, g = make_case(oracle, 80000, 300, enzyme=3, seed=5)
    gpu, ora, tm = run_both(capi, oracle, odb, t, p, g, 3, 4, 2000, plan=plan)
    assert tm.prefix_bases == plan[0]
    assert_same_hits(gpu, ora)


def dense_case(oracle, n_random=60000, n_guides=300, n_dense=40, variants=80, seed=21):
    """random database + a dense neighbourhood (<= 3 substitutions) around the first n_dense guides, so that those
    guides collect many hits and the cut-off rule is exercised"""
    g = synth.make_guides(n_guides, seed=synth.GUIDE_SEED + seed)
    db = synth.make_database(n_random, seed=synth.DB_SEED + seed, plant_guides=g, with_positions=False)
    rng = np.random.default_rng(seed)
    mers = set(int(x) for x in ((db["targets"] >> 6) & synth.MASK40))
    for gi in range(n_dense):
        base = (int(g[gi]) >> 6) & synth.MASK40
        for _ in range(variants):
            v = base
            for _ in range(int(rng.integers(1, 4))):
                v ^= int(rng.integers(1, 4)) << (2 * int(rng.integers(0, 20)))
            mers.add(v)
    mers = np.array(sorted(mers), dtype=np.uint64)
    counts = rng.integers(1, 6, size=len(mers)).astype(np.uint64)
    targets = (mers << np.uint64(6)) | np.uint64(0b101010) | (counts << np.uint64(48))
    positions = rng.integers(0, 1 << 27, size=int(counts.sum()), dtype=np.uint64) | (np.uint64(23) << np.uint64(52)) | (np.uint64(3) << np.uint64(32))
    odb = oracle.db_from_sorted(3, targets, positions, contigs=synth.CONTIGS_24)
    return odb, targets, positions, synth.as_u64(g)


@pytest.mark.parametrize("max_ot", [0, 1, 5, 37, 2000])
def test_ordered_cutoff(capi, oracle, max_ot):
    """CRISPRSiteOT.addOT/full (CRISPRSiteOT.scala:39-46): keep while the running position total is < max"""
    odb, t, p, g = dense_case(oracle)
    gpu, ora, _ = run_both(capi, oracle, odb, t, p, g, 3, 4, max_ot)
    assert_same_hits(gpu, ora)
    if 0 < max_ot < 2000:
        assert ora.full.any() and not ora.full.all()
        # the last retained hit may overshoot the limit (addOT adds the whole position list, CRISPRSiteOT.scala:45)
        assert (ora.current_total[ora.full] >= max_ot).all()
    assert_same_scores(oracle, 3, g, gpu, ora)


@pytest.mark.parametrize("enzyme", [1, 4, 5, 6])
def test_other_enzymes(capi, oracle, enzyme):
    """Cpf1 (5' PAM, bases 4..23 compared), NAG, and the 19-mer packs (StandardScanParameters.scala:90-215)"""
    rng = np.random.default_rng(enzyme)
    L = {1: 24, 4: 23, 5: 22, 6: 22}[enzyme]
    n = 40000
    raw = rng.integers(0, 1 << (2 * L), size=n, dtype=np.uint64)
    raw = np.unique(raw)
    counts = rng.integers(1, 4, size=len(raw)).astype(np.uint64)
    targets = raw | (counts << np.uint64(48))
    if enzyme == 1:  # database order of a 5'-PAM enzyme: bin = bases 4..10, then the full string
        binkey = (raw >> np.uint64(2 * (24 - 11))) & np.uint64(0x3FFF)
        order = np.lexsort((raw, binkey))
        targets = targets[order]
        raw = raw[order]
    positions = rng.integers(0, 1 << 27, size=int(counts.sum()), dtype=np.uint64) | (np.uint64(L) << np.uint64(52)) | (np.uint64(1) << np.uint64(32))
    if enzyme == 1:
        counts = (targets >> np.uint64(48))
    odb = oracle.db_from_sorted(enzyme, targets, positions, contigs=["c1"])
    # guides = mutated database members so that hits exist
    pick = rng.integers(0, len(raw), size=200)
    guides = raw[pick].copy()
    for k in range(len(guides)):
        for _ in range(rng.integers(0, 5)):
            guides[k] ^= np.uint64(int(rng.integers(1, 4)) << (2 * int(rng.integers(0, L))))
    guides |= np.uint64(1) << np.uint64(48)
    gpu, ora, _ = run_both(capi, oracle, odb, targets, positions, guides, enzyme, 3, 2000)
    assert_same_hits(gpu, ora)
    assert gpu.n_hits >= 50
    assert_same_scores(oracle, enzyme, guides, gpu, ora, jost=True)


@pytest.mark.parametrize("max_mm", [11, 12, 20, 25])
def test_huge_mismatch_budgets(capi, oracle, max_mm):
    """maxMismatch >= the guide length: every target is a hit; exercises the single-image fallback plan and the
    sentinel-free compare variant (max_mm >= 12), with the cut-off deciding what is kept"""
    odb, t, p, g = make_case(oracle, 3000, 12, enzyme=3, seed=3)
    gpu, ora, tm = run_both(capi, oracle, odb, t, p, g, 3, max_mm, 150)
    assert_same_hits(gpu, ora)
    assert ora.full.any() and (ora.full.all() or max_mm < 20)
    gpu, ora, tm = run_both(capi, oracle, odb, t, p, g[:3], 3, max_mm, 10 ** 7)
    assert_same_hits(gpu, ora)
    if max_mm >= 20:
        assert gpu.n_hits == 3 * len(t)
    assert_same_scores(oracle, 3, g[:3], gpu, ora)


def test_guide_batches(capi, oracle, monkeypatch):
    """the candidate lists are built per guide batch; any batch size gives the same result"""
    odb, t, p, g = dense_case(oracle, seed=9)
    ora = odb.discover(g, 4, 60)
    for batch in ("1000000", "97", "7"):
        monkeypatch.setenv("FFH_MAX_GUIDE_BATCH", batch)
        with capi.Context(3) as ctx:
            ctx.load_soa(t, p)
            gpu = ctx.discover(g, 4, 60)
            assert ctx.timings().compare_launches == max(1, -(-len(g) // int(batch)))
        assert_same_hits(gpu, ora)
        assert_same_scores(oracle, 3, g, gpu, ora)


def test_compare_grid_sizes(capi, oracle, monkeypatch):
    """the work items are dealt to the waves of the compare launch round-robin; the result must not depend on how many there are
    (one block: every wave walks thousands of items through the software pipeline; far more waves than items: most leave at once)"""
    odb, t, p, g = dense_case(oracle, seed=10)
    ora = odb.discover(g, 4, 60)
    for grid in ("1", "3", "64", "100000"):
        monkeypatch.setenv("FFH_COMPARE_GRID", grid)
        with capi.Context(3) as ctx:
            ctx.load_soa(t, p)
            gpu = ctx.discover(g, 4, 60)
        assert_same_hits(gpu, ora)
        assert_same_scores(oracle, 3, g, gpu, ora)


def test_edge_cases(capi, oracle):
    odb, t, p, g = make_case(oracle, 5000, 50, enzyme=3, seed=2)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        r = ctx.discover(np.zeros(0, dtype=np.uint64), 4, 2000)           # no guides
        assert r.n_guides == 0 and r.n_hits == 0
        r = ctx.discover(np.zeros(0, dtype=np.uint64), 4, 2000, summaries_only=True)
        assert r.n_guides == 0 and r.n_hits == 0 and len(r.summaries) == 0
        far = np.array([oracle.encode("ACGT" * 5 + "AGG")], dtype=np.uint64)  # a guide without any hit
        r = ctx.discover(far, 0, 2000)
        assert r.n_guides == 1
        o = odb.discover(far, 0, 2000)
        assert_same_hits(r, o)
        dup = np.concatenate([g[:10], g[:10]])                              # duplicated guides give separate rows (quirk 11)
        r = ctx.discover(dup, 4, 2000)
        assert_same_hits(r, odb.discover(dup, 4, 2000))
        assert np.array_equal(r.hits(0), r.hits(10))
    with capi.Context(3) as ctx:                                            # empty database
        ctx.load_soa(np.zeros(0, np.uint64), np.zeros(0, np.uint64))
        r = ctx.discover(g, 4, 2000)
        assert r.n_hits == 0 and not r.summaries["overflow"].any()
    with capi.Context(3) as ctx:                                            # malformed input is refused
        bad = t.copy()
        bad[3] &= np.uint64((1 << 48) - 1)                                  # count 0
        with pytest.raises(capi.FlashFryHipError):
            ctx.load_soa(bad, p)
        with pytest.raises(capi.FlashFryHipError):
            ctx.load_blocks(np.array([7, 1, 2], dtype=np.int64), np.array([0, 3], dtype=np.uint64))  # unknown block type
        with pytest.raises(capi.FlashFryHipError):
            ctx.discover(g, 4, 2000)                                       # no database loaded


def test_hit_buffer_growth_and_heavy_guides(capi, oracle):
    """guides with very many hits: poly-A database, max_mm large, staging buffer must regrow; cut-off still exact"""
    rng = np.random.default_rng(3)
    base = oracle.encode("A" * 20 + "AGG") & ((1 << 46) - 1)
    muts = set()
    while len(muts) < 30000:
        v = base
        for _ in range(rng.integers(0, 5)):
            v ^= int(rng.integers(1, 4)) << (2 * int(rng.integers(3, 23)))
        muts.add(v)
    raw = np.array(sorted(muts), dtype=np.uint64)
    targets = raw | (np.uint64(1) << np.uint64(48))
    positions = np.arange(len(raw), dtype=np.uint64) | (np.uint64(23) << np.uint64(52)) | (np.uint64(1) << np.uint64(32))
    odb = oracle.db_from_sorted(3, targets, positions, contigs=["c1"])
    guides = np.array([oracle.encode("A" * 20 + "TGG"), oracle.encode("A" * 19 + "C" + "TGG"), oracle.encode("ACGT" * 5 + "TGG")], dtype=np.uint64)
    for max_ot in (2000, 10 ** 6):
        gpu, ora, _ = run_both(capi, oracle, odb, targets, positions, guides, 3, 6, max_ot)
        assert_same_hits(gpu, ora)
    assert gpu.n_hits >= 30000


def test_two_shards_with_ordered_cutoff(capi, oracle):
    """bins split over two contexts; totals of the first shard shift the cut-off of the second"""
    odb, t, p, g = dense_case(oracle, seed=31)
    max_ot = 40
    ora = odb.discover(g, 5, max_ot)
    assert ora.full.any() and not ora.full.all()
    cut = len(t) // 2
    poff = np.concatenate([[0], np.cumsum(t >> np.uint64(48))]).astype(np.int64)
    parts = [(t[:cut], p[:poff[cut]]), (t[cut:], p[poff[cut]:])]
    ctxs = [capi.Context(3), capi.Context(3)]
    try:
        for c, (tt, pp) in zip(ctxs, parts):
            c.load_soa(tt, pp)
            c.scan(g, 5)
        totals = [c.shard_totals(max_ot) for c in ctxs]
        prior = [np.zeros(len(g), np.uint32), totals[0]]
        res = [c.finalize(max_ot, prior_totals=pr) for c, pr in zip(ctxs, prior)]
        only = [c.finalize(max_ot, prior_totals=pr, summaries_only=True) for c, pr in zip(ctxs, prior)]   # the fused epilogue continues the cut-off too
        assert all(o.summaries.tobytes() == r.summaries.tobytes() for o, r in zip(only, res))
    finally:
        for c in ctxs:
            c.close()
    n_hits = res[0].summaries["n_hits"].astype(np.int64) + res[1].summaries["n_hits"]
    assert np.array_equal(n_hits, np.diff(ora.guide_offsets.astype(np.int64)))
    for gi in range(len(g)):
        assert np.array_equal(np.concatenate([res[0].hits(gi), res[1].hits(gi)]), ora.hits(gi))
    tot = res[0].summaries["ot_count"].astype(np.int64) + res[1].summaries["ot_count"]
    assert np.array_equal(tot, ora.current_total)
    assert np.array_equal((res[0].summaries["overflow"] | res[1].summaries["overflow"]).astype(bool), ora.full)


def test_database_file_roundtrip(capi, oracle, tmp_path):
    """a database written in the reference's on-disk format (.header + BGZF) loads into the HIP path, whole and by bin range"""
    odb, t, p, g = make_case(oracle, 50000, 100, enzyme=3, seed=41, max_linear=3)
    path = str(tmp_path / "synthetic_db")
    odb.write(path)
    ora = odb.discover(g, 4, 2000)
    with capi.Context(3) as ctx:
        ctx.open(path)
        info = ctx.info()
        assert info.n_targets == len(t) and info.n_positions == len(p) and info.n_bins == 16384
        assert ctx.contigs() == synth.CONTIGS_24
        gpu = ctx.discover(g, 4, 2000)
    assert_same_hits(gpu, ora)
    with capi.Context(3) as c0, capi.Context(3) as c1:
        c0.open(path, 0, 8192)
        c1.open(path, 8192, 0)
        assert c0.info().n_targets + c1.info().n_targets == len(t)
        c0.scan(g, 4); c1.scan(g, 4)
        r0 = c0.finalize(2000)
        r1 = c1.finalize(2000, prior_totals=c0.shard_totals(2000))
    for gi in range(len(g)):
        assert np.array_equal(np.concatenate([r0.hits(gi), r1.hits(gi)]), ora.hits(gi))


def _rewrite_bgzf(src, dst, member_bytes, level, strategy):
    """the same database body re-compressed into BGZF members of `member_bytes` payload with the given zlib settings"""
    import gzip
    import zlib
    raw = gzip.open(src).read()
    hdr = open(src + ".header").read().split("\n")
    nbins = int(hdr[3])
    members, coff, off = [], [0], 0
    for a in range(0, len(raw), member_bytes):
        chunk = raw[a:a + member_bytes]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        data = co.compress(chunk) + co.flush()
        total = 18 + len(data) + 8
        assert total <= 65536
        members.append(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + (total - 1).to_bytes(2, "little") + data +
                       (zlib.crc32(chunk) & 0xFFFFFFFF).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))
        off += total
        coff.append(off)
    eof = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    with open(dst, "wb") as f:
        f.write(b"".join(members) + eof)
    lin = 0
    for b in range(nbins):
        name, rest = hdr[4 + b].split("=")
        _, nbytes, nt = rest.split(",")
        hdr[4 + b] = "%s=%d,%s,%s" % (name, (coff[lin // member_bytes] << 16) | (lin % member_bytes), nbytes, nt)
        lin += int(nbytes)
    assert lin == len(raw)
    with open(dst + ".header", "w") as f:
        f.write("\n".join(hdr))


@pytest.mark.parametrize("member_bytes,level,strategy", [(65280, 5, "default"), (65000, 0, "default"), (4096, 9, "default"), (32768, 6, "fixed"),
                                                         (65280, 6, "huffman_only"), (8192, 1, "rle"), (65280, 9, "filtered")])
def test_device_inflate_handles_every_deflate_block_type(capi, oracle, tmp_path, monkeypatch, member_bytes, level, strategy):
    """the BGZF members are inflated on the device (ffh_inflate.hpp; FFH_INFLATE=device forces it for a body this small, which the
    host threads take by default since round 5): stored, fixed-Huffman and dynamic-Huffman blocks, short and long matches, small and
    full-size members; the host-thread inflate (FFH_INFLATE=host, and the default here) must give the same database"""
    import zlib
    odb, t, p, g = make_case(oracle, 40000, 60, enzyme=3, seed=77, max_linear=500)
    src, dst = str(tmp_path / "src_db"), str(tmp_path / "re_db")
    odb.write(src)
    strat = {"default": zlib.Z_DEFAULT_STRATEGY, "fixed": zlib.Z_FIXED, "huffman_only": zlib.Z_HUFFMAN_ONLY, "rle": zlib.Z_RLE, "filtered": zlib.Z_FILTERED}[strategy]
    _rewrite_bgzf(src, dst, member_bytes, level, strat)
    ora = odb.discover(g, 4, 2000)
    for where in ("device", "host", "default", "pipeline"):
        monkeypatch.delenv("FFH_INFLATE", raising=False)
        monkeypatch.delenv("FFH_LOAD_PIPELINE", raising=False)
        if where in ("device", "host"):
            monkeypatch.setenv("FFH_INFLATE", where)
        if where == "pipeline":
            monkeypatch.setenv("FFH_LOAD_PIPELINE", "1")        # the large-body path (page-locked arena, copy streams, device inflate)
        with capi.Context(3) as ctx:
            ctx.open(dst)
            st = ctx.load_stats()
            assert (st.device_inflate_ms > 0) == (where in ("device", "pipeline"))
            assert (ctx.info().n_targets, ctx.info().n_positions) == (len(t), len(p))
            assert_same_hits(ctx.discover(g, 4, 2000), ora)
            c2 = capi.Context(3)
            c2.open(dst, 5000, 9000)                      # a bin range that starts and ends inside members
            n_mid = c2.info().n_targets
            c2.close()
        assert 0 < n_mid < len(t)
    # a flipped payload bit must be caught by the CRC check, on the device and on the host threads
    monkeypatch.delenv("FFH_LOAD_PIPELINE", raising=False)
    blob = bytearray(open(dst, "rb").read())
    blob[len(blob) // 2] ^= 0x10
    open(dst, "wb").write(bytes(blob))
    for where in ("device", None):
        if where:
            monkeypatch.setenv("FFH_INFLATE", where)
        else:
            monkeypatch.delenv("FFH_INFLATE", raising=False)
        with capi.Context(3) as ctx:
            with pytest.raises(capi.FlashFryHipError, match="inflate / crc failure|BGZF"):
                ctx.open(dst)


def test_device_resident_exchange_entry_points(capi, oracle):
    """ffh_shard_totals_device / FFH_FINALIZE_PRIOR_ON_DEVICE / ffh_summaries_to_device (what bench.py uses for N > 1) against the
    host-pointer forms, with two shards on the one GPU; then flashfry_amd.dist.DeviceExchange over a 1-rank RCCL group"""
    import socket
    import torch
    import torch.distributed as dist
    from flashfry_amd import dist as ffdist
    odb, t, p, g = dense_case(oracle, seed=12)
    cut = len(t) // 2
    counts = (t >> np.uint64(48)).astype(np.int64)
    pcut = int(counts[:cut].sum())
    dev = torch.device("cuda", 0)
    max_ot = 60
    with capi.Context(3) as c0, capi.Context(3) as c1:
        c0.load_soa(t[:cut], p[:pcut])
        c1.load_soa(t[cut:], p[pcut:])
        c0.scan(g, 5)
        c1.scan(g, 5)
        tot_dev = torch.zeros(len(g), dtype=torch.int32, device=dev)
        c0.shard_totals_device(tot_dev.data_ptr(), max_ot)
        host_tot = c0.shard_totals(max_ot)
        assert np.array_equal(tot_dev.cpu().numpy().astype(np.uint32), host_tot)
        a = c1.finalize(max_ot, prior_totals=host_tot, summaries_only=True)
        b = c1.finalize_device_prior(max_ot, tot_dev.data_ptr(), summaries_only=True)
        assert a.summaries.tobytes() == b.summaries.tobytes() and a.summaries["overflow"].any()
        buf = torch.zeros(len(g) * capi.SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        c1.summaries_to_device(buf.data_ptr())
        assert buf.cpu().numpy().tobytes() == b.summaries.tobytes()
    # the pack / mask / unpack kernels around the three collectives, with the collectives of a 2-rank group done by hand
    with capi.Context(3) as c0, capi.Context(3) as c1:
        c0.load_soa(t[:cut], p[:pcut])
        c1.load_soa(t[cut:], p[pcut:])
        c0.scan(g, 5)
        c1.scan(g, 5)
        s0 = c0.finalize(max_ot, summaries_only=True, jost=True).summaries.copy()
        s1 = c1.finalize(max_ot, prior_totals=c0.shard_totals(max_ot), summaries_only=True, jost=True).summaries.copy()
        n, isz = len(g), capi.SUMMARY_DTYPE.itemsize
        bufs, mxs, sums, fsums = [], [], [], []
        for c in (c0, c1):
            b = torch.zeros(n * isz, dtype=torch.uint8, device=dev)
            c.summaries_to_device(b.data_ptr())
            mx, sm, fs = torch.zeros(n * 4, dtype=torch.float64, device=dev), torch.zeros(n * 10, dtype=torch.int32, device=dev), torch.zeros(n * 3, dtype=torch.float64, device=dev)
            c.exchange_pack(b.data_ptr(), n, mx.data_ptr(), sm.data_ptr(), fs.data_ptr())
            bufs.append(b); mxs.append(mx); sums.append(sm); fsums.append(fs)
        mx = torch.maximum(mxs[0], mxs[1])                                   # all-reduce MAX
        for c, b, sm in zip((c0, c1), bufs, sums):
            c.exchange_mask(b.data_ptr(), n, mx.data_ptr(), sm.data_ptr())
        total = sums[0] + sums[1]                                            # all-reduce SUM
        gathered = torch.cat(fsums)                                          # all-gather, rank order
        c0.exchange_unpack(bufs[0].data_ptr(), n, mx.data_ptr(), total.data_ptr(), gathered.data_ptr(), 2)
        got = bufs[0].cpu().numpy().view(capi.SUMMARY_DTYPE)
    want = s0.copy()
    for f in ("n_hits", "ot_count", "hist", "in_genome", "n_scored"):
        want[f] = s0[f] + s1[f]
    for f in ("overflow", "cfd_max", "jost_max"):
        want[f] = np.maximum(s0[f], s1[f])
    want["closest"] = np.minimum(s0["closest"], s1["closest"])
    want["closest_count"] = np.where(s0["closest"] == want["closest"], s0["closest_count"], 0) + np.where(s1["closest"] == want["closest"], s1["closest_count"], 0)
    for f in ("cfd_sum", "hsu_sum", "jost_sum"):
        want[f] = s0[f] + s1[f]
    bad = [i for i in range(n) if got[i].tobytes() != want[i].tobytes()]
    assert not bad, (bad[:3], got[bad[0]], want[bad[0]], s0[bad[0]], s1[bad[0]])
    assert (want["closest"] != 0xFFFFFFFF).any() and (s1["n_hits"] > 0).any()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0, device_id=dev)
    try:
        with capi.Context(3) as ctx:
            ctx.load_soa(t, p)
            ctx.scan(g, 5)
            want = ctx.finalize(max_ot, summaries_only=True, jost=True).summaries.copy()
            ex = ffdist.DeviceExchange(len(g), dev)
            ex.step(ctx, max_ot, jost=True)
            assert ex.summaries_numpy().tobytes() == want.tobytes()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("on_caller_stream", [False, "default", "side"])
def test_one_pass_shard_finalize_equals_the_two_pass_exchange(capi, oracle, on_caller_stream):
    """ffh_finalize_shard / ffh_exchange_prior / ffh_finalize_shard_fixup (one aggregation pass per shard; only the guides whose
    cut-off the earlier shards move are aggregated again) against ffh_shard_totals + ffh_finalize(prior): three shards on the one
    GPU, a maximumOffTargets small enough that the limit is reached in the first, the second, the third shard or never"""
    import contextlib
    import torch
    odb, t, p, g = dense_case(oracle, seed=21)
    side = torch.cuda.Stream() if on_caller_stream == "side" else None   # a stream of the caller's other than the default one
    scope = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
    counts = (t >> np.uint64(48)).astype(np.int64)
    cuts = [0, len(t) // 3, 2 * len(t) // 3, len(t)]
    pcuts = [int(counts[:c].sum()) for c in cuts]
    dev = torch.device("cuda", 0)
    n, isz = len(g), capi.SUMMARY_DTYPE.itemsize
    for max_ot in (25, 60, 2000):
        ctxs = [capi.Context(3) for _ in range(3)]
        try:
            want, prior_want, tots = [], [], []
            for r, c in enumerate(ctxs):
                c.load_soa(t[cuts[r]:cuts[r + 1]], p[pcuts[r]:pcuts[r + 1]])
                c.scan(g, 5)
                prior = np.minimum(np.sum(tots, axis=0, dtype=np.int64), max_ot).astype(np.uint32) if tots else None
                prior_want.append(prior if prior is not None else np.zeros(n, np.uint32))
                want.append(c.finalize(max_ot, prior_totals=prior, summaries_only=True, jost=True).summaries.copy())
                tots.append(c.shard_totals(max_ot).astype(np.int64))
            with scope():
                if on_caller_stream:
                    for c in ctxs:
                        c.use_stream(torch.cuda.current_stream().cuda_stream)
                summ = [torch.zeros(n * isz, dtype=torch.uint8, device=dev) for _ in ctxs]
                totals = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in ctxs]
                for c, sm, tt in zip(ctxs, summ, totals):
                    c.finalize_shard(max_ot, sm.data_ptr(), tt.data_ptr(), jost=True)
                all_totals = torch.cat(totals)                                        # the all-gather
                redone = 0
                for r, (c, sm, tt) in enumerate(zip(ctxs, summ, totals)):
                    assert np.array_equal(tt.cpu().numpy().astype(np.int64), tots[r])
                    prior = torch.zeros(n, dtype=torch.int32, device=dev)
                    c.exchange_prior(all_totals.data_ptr(), n, r, max_ot, prior.data_ptr())
                    assert np.array_equal(prior.cpu().numpy().astype(np.uint32), prior_want[r])
                    before = sm.cpu().numpy().tobytes()
                    c.finalize_shard_fixup(max_ot, prior.data_ptr(), tt.data_ptr(), sm.data_ptr(), jost=True)
                    after = sm.cpu().numpy()
                    redone += before != after.tobytes()
                    got = after.view(capi.SUMMARY_DTYPE)
                    bad = [i for i in range(n) if got[i].tobytes() != want[r][i].tobytes()]
                    assert not bad, (max_ot, r, bad[:3], got[bad[0]], want[r][bad[0]])
                if on_caller_stream:
                    for c in ctxs:
                        c.use_stream(0, on=False)   # back on their own streams before the caller's stream goes away
            if max_ot < 2000:
                assert redone >= 1 and want[2]["overflow"].any() and not want[2]["overflow"].all()
            else:
                assert redone == 0
        finally:
            for c in ctxs:
                c.close()


def test_cas12a_bulge_search_matches_the_brute_force_specification(capi, oracle):
    """config C5 (Cas12a TTTV, mismatches + one bulge; the reference has no bulge search: parity is against this repository's own
    specification, restated with strings in the oracle): every (guide, target) pair of a small database, all three alignment kinds"""
    rng = np.random.default_rng(99)
    L = 24
    pam = int(oracle.encode("TTTA" + "A" * 20)) >> 40 << 40                     # TTTA....: the PAM bits of a 24-mer
    raw = np.unique(rng.integers(0, 1 << 40, size=12000, dtype=np.uint64))
    pam_n = rng.integers(0, 4, size=len(raw)).astype(np.uint64)                 # TTTN: the fourth PAM base varies
    seq = (np.uint64(0b111111) << np.uint64(42)) | (pam_n << np.uint64(40)) | raw
    guides = []
    planted = []
    for k in range(42):                                                          # guides whose bulged / mismatched copies are planted
        g = int(rng.integers(0, 1 << 40))
        guides.append(g | (0b11111100 << 40) | (1 << 48))
        bases = [(g >> (2 * (19 - i))) & 3 for i in range(20)]
        kind, pos = k % 3, int(rng.integers(1, 19))
        if kind == 1:    # RNA bulge: the target lacks guide base `pos`; its last base is free
            tb = bases[:pos] + bases[pos + 1:] + [int(rng.integers(0, 4))]
        elif kind == 2:  # DNA bulge: the target has an extra base at `pos`
            tb = (bases[:pos] + [int(rng.integers(0, 4))] + bases[pos:])[:20]
        else:
            tb = list(bases)
        for _ in range(int(rng.integers(0, 4))):                                 # up to 3 mismatches on top
            tb[int(rng.integers(0, 20))] = int(rng.integers(0, 4))
        v = 0
        for b in tb:
            v = (v << 2) | b
        planted.append(v | (0b111111 << 42) | (int(rng.integers(0, 4)) << 40))
    seq = np.unique(np.concatenate([seq, np.array(planted, dtype=np.uint64)]))
    binkey = (seq >> np.uint64(2 * (24 - 11))) & np.uint64(0x3FFF)                # database order of a 5'-PAM enzyme
    seq = seq[np.lexsort((seq, binkey))]
    counts = rng.integers(1, 3, size=len(seq)).astype(np.uint64)
    targets = seq | (counts << np.uint64(48))
    positions = rng.integers(0, 1 << 27, size=int(counts.sum()), dtype=np.uint64)
    guides = np.array(guides, dtype=np.uint64)
    pk = oracle.pack(1)
    ty, ps = capi.C.c_int(), capi.C.c_int()
    for max_mm, max_bulge, tttv in ((3, 1, False), (2, 1, True), (3, 0, False)):
        with capi.Context(1) as ctx:
            ctx.load_soa(targets, positions)
            res = ctx.discover_bulge(guides, max_mm, max_bulge, tttv=tttv)
        want = []
        for gi, g in enumerate(guides):
            for t in targets:
                if tttv and ((int(t) >> 40) & 3) == 3:
                    continue
                mm = oracle.lib.ffo_bulge_align(pk, int(g), int(t), max_bulge, capi.C.byref(ty), capi.C.byref(ps))
                if mm <= max_mm:
                    want.append((gi, int(t), mm, ty.value, ps.value))
        got = []
        for gi in range(len(guides)):
            a, b = int(res.guide_offsets[gi]), int(res.guide_offsets[gi + 1])
            got += [(gi, int(res.hit_targets[h]), int(res.hit_mismatches[h]), int(res.hit_bulge_type[h]), int(res.hit_bulge_position[h])) for h in range(a, b)]
        assert got == want, (max_mm, max_bulge, tttv, len(got), len(want))
        kinds = {w[3] for w in want}
        assert kinds == ({0, 1, 2} if max_bulge else {0}) and len(want) >= (30 if max_bulge and not tttv else 10)
    with capi.Context(3) as ctx:                                                  # specified for Cas12a only
        ctx.load_soa(np.zeros(0, np.uint64), np.zeros(0, np.uint64))
        with pytest.raises(capi.FlashFryHipError, match="Cas12a"):
            ctx.discover_bulge(guides, 3, 1)


def test_hit_lists_without_positions(capi, oracle):
    """FFH_FINALIZE_NO_POSITIONS: same lists, scores and aggregates, no position arrays (the count of a hit is in its target long)"""
    odb, t, p, g = dense_case(oracle, seed=5)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        full = ctx.discover(g, 4, 60, jost=True)
        lean = ctx.finalize(60, jost=True, positions=False)
    assert lean.positions is None and lean.pos_offsets is None and lean.n_positions == 0
    assert np.array_equal(lean.guide_offsets, full.guide_offsets) and np.array_equal(lean.hit_targets, full.hit_targets)
    assert np.array_equal(lean.hit_mismatches, full.hit_mismatches)
    assert lean.hit_cfd.tobytes() == full.hit_cfd.tobytes() and lean.summaries.tobytes() == full.summaries.tobytes()
    assert np.array_equal((lean.hit_targets >> np.uint64(48)).astype(np.int64), np.diff(full.pos_offsets.astype(np.int64)))


def test_hit_lists_without_per_hit_scores(capi, oracle):
    """FFH_FINALIZE_NO_HIT_SCORES (what the CLI's discover and the JNI binding pass): sequences, mismatches, positions and aggregates
    unchanged, no pam*cfd array; pos_offsets (never copied from the device, folded from the counts on first use) against the oracle"""
    odb, t, p, g = dense_case(oracle, seed=6)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        full = ctx.discover(g, 4, 60, jost=True)
        lean = ctx.finalize(60, jost=True, hit_scores=False)
        bare = ctx.finalize(60, jost=True, hit_scores=False, positions=False)
    ora = odb.discover(g, 4, 60)
    assert lean.hit_cfd is None and bare.hit_cfd is None and bare.positions is None
    for r in (lean, bare):
        assert np.array_equal(r.guide_offsets, full.guide_offsets) and np.array_equal(r.hit_targets, full.hit_targets)
        assert np.array_equal(r.hit_mismatches, full.hit_mismatches) and r.summaries.tobytes() == full.summaries.tobytes()
    assert np.array_equal(lean.pos_offsets, full.pos_offsets) and np.array_equal(lean.positions, full.positions)
    assert_same_hits(full, ora)
    assert_same_hits(lean, ora)


def test_position_offsets_of_a_large_result_are_folded_by_several_host_threads(capi, oracle):
    """more than 262 144 retained hits: ffh_result_pos_offsets splits the fold over host threads; equal to numpy's cumulative sum"""
    odb, t, p, g = dense_case(oracle, n_random=50000, n_guides=400, n_dense=400, variants=1600, seed=77)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        r = ctx.discover(g, 4, 100000, hit_scores=False)
    assert r.n_hits > 300_000
    want = np.concatenate([[0], np.cumsum((r.hit_targets >> np.uint64(48)).astype(np.uint64))]).astype(np.uint64)
    assert np.array_equal(r.pos_offsets, want) and int(want[-1]) == r.n_positions == len(r.positions)
    ora = odb.discover(g[:50], 4, 100000)
    a = int(r.guide_offsets[50])
    assert np.array_equal(r.positions[:int(r.pos_offsets[a])], ora.positions)


def _bulge_by_prefix_suffix_sums(g_bases, t_bases, max_bulge):
    """A second, independently written checker of the bulge specification: instead of re-counting every
    alignment base by base (the oracle's triple loop), the mismatch indicators of the three diagonals of the alignment matrix
    -- guide i against target i, i - 1, i + 1 -- are cumulated once; an alignment with one gap is a prefix of the main diagonal
    plus a suffix of an off diagonal, so every candidate is two table look-ups (the DP over <= 1 gap, unrolled).  Vectorised over
    all targets.  Returns (best mismatches, type 0 none / 1 RNA / 2 DNA, position)."""
    n = 20
    T = t_bases.shape[0]
    d0 = (t_bases != g_bases[None, :]).astype(np.int32)                       # g_i vs t_i
    dm = (t_bases[:, :n - 1] != g_bases[None, 1:]).astype(np.int32)           # g_i vs t_{i-1}, i = 1..19  (index i - 1)
    dp = (t_bases[:, 1:] != g_bases[None, :n - 1]).astype(np.int32)           # g_i vs t_{i+1}, i = 0..18  (index i)
    pre0 = np.concatenate([np.zeros((T, 1), np.int32), np.cumsum(d0, 1)], 1)  # pre0[:, k] = mismatches of pairs 0 .. k-1
    sm = np.concatenate([np.cumsum(dm[:, ::-1], 1)[:, ::-1], np.zeros((T, 1), np.int32)], 1)   # sm[:, j] = sum dm[j:]
    sp = np.concatenate([np.cumsum(dp[:, ::-1], 1)[:, ::-1], np.zeros((T, 1), np.int32)], 1)   # sp[:, j] = sum dp[j:]
    best = pre0[:, n].copy()
    btype = np.zeros(T, np.int32)
    bpos = np.zeros(T, np.int32)
    if max_bulge:
        for kind in (1, 2):                                                    # RNA first: it wins ties against DNA
            for k in range(1, n - 1):
                # RNA bulge at k: guide base k unpaired, g_i ~ t_{i-1} for i > k  -> dm indices k .. 18
                # DNA bulge at k: target base k unpaired, g_i ~ t_{i+1} for k <= i <= 18 -> dp indices k .. 18
                mm = pre0[:, k] + (sm[:, k] if kind == 1 else sp[:, k])
                better = mm < best
                best = np.where(better, mm, best)
                btype = np.where(better, kind, btype)
                bpos = np.where(better, k, bpos)
    return best, btype, bpos


def test_cas12a_bulge_search_against_a_second_independent_checker(capi, oracle):
    """the same specification checked by differently structured code (cumulated diagonals instead of per-alignment loops), on a larger
    database than the oracle's pair-by-pair loop can cover: 60 000 targets x 60 guides = 3.6e6 pairs, three parameter sets"""
    rng = np.random.default_rng(2024)
    raw = np.unique(rng.integers(0, 1 << 40, size=60000, dtype=np.uint64))
    guides40 = rng.integers(0, 1 << 40, size=60, dtype=np.uint64)
    extra = []
    for g in guides40:                                                          # near-copies so that every alignment kind occurs
        bases = [(int(g) >> (2 * (19 - i))) & 3 for i in range(20)]
        for _ in range(25):
            kind, pos = int(rng.integers(0, 3)), int(rng.integers(1, 19))
            tb = list(bases) if kind == 0 else (bases[:pos] + bases[pos + 1:] + [int(rng.integers(0, 4))] if kind == 1 else (bases[:pos] + [int(rng.integers(0, 4))] + bases[pos:])[:20])
            for _ in range(int(rng.integers(0, 4))):
                tb[int(rng.integers(0, 20))] = int(rng.integers(0, 4))
            v = 0
            for b in tb:
                v = (v << 2) | b
            extra.append(v)
    raw = np.unique(np.concatenate([raw, np.array(extra, dtype=np.uint64)]))
    pam_n = rng.integers(0, 4, size=len(raw)).astype(np.uint64)
    seq = (np.uint64(0b111111) << np.uint64(42)) | (pam_n << np.uint64(40)) | raw
    binkey = (seq >> np.uint64(2 * (24 - 11))) & np.uint64(0x3FFF)
    seq = seq[np.lexsort((seq, binkey))]
    targets = seq | (np.uint64(1) << np.uint64(48))
    positions = rng.integers(0, 1 << 27, size=len(seq), dtype=np.uint64)
    guides = guides40 | np.uint64(0b11111100 << 40) | np.uint64(1 << 48)
    shifts = (2 * (19 - np.arange(20))).astype(np.uint64)
    t_bases = ((seq[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.int8)
    for max_mm, max_bulge, tttv in ((3, 1, False), (2, 1, True), (4, 0, False)):
        with capi.Context(1) as ctx:
            ctx.load_soa(targets, positions)
            res = ctx.discover_bulge(guides, max_mm, max_bulge, tttv=tttv)
            bf = ctx.discover_bulge(guides, max_mm, max_bulge, tttv=tttv, brute_force=True)
        for name in ("guide_offsets", "hit_targets", "hit_mismatches", "hit_bulge_type", "hit_bulge_position"):
            assert np.array_equal(getattr(res, name), getattr(bf, name)), name      # seeded candidate search == every pair
        n_checked = 0
        for gi, g in enumerate(guides40):
            g_bases = ((np.uint64(g) >> shifts) & np.uint64(3)).astype(np.int8)
            best, btype, bpos = _bulge_by_prefix_suffix_sums(g_bases, t_bases, max_bulge)
            keep = best <= max_mm
            if tttv:
                keep &= ((seq >> np.uint64(40)) & np.uint64(3)) != 3
            idx = np.nonzero(keep)[0]
            a, b = int(res.guide_offsets[gi]), int(res.guide_offsets[gi + 1])
            assert np.array_equal(res.hit_targets[a:b], targets[idx]), (gi, max_mm, max_bulge)
            assert np.array_equal(res.hit_mismatches[a:b], best[idx].astype(np.uint8))
            assert np.array_equal(res.hit_bulge_type[a:b], btype[idx].astype(np.uint8))
            assert np.array_equal(res.hit_bulge_position[a:b], bpos[idx].astype(np.uint8))
            n_checked += len(idx)
        assert n_checked >= 60


def _cas12a_database(rng, n_random, guides40, copies):
    """TTTN 24-mers in the database order of a 5'-PAM enzyme: random protospacers + near-copies (substitutions, RNA / DNA bulges) of the guides"""
    raw = rng.integers(0, 1 << 40, size=n_random, dtype=np.uint64)
    extra = []
    for g in guides40:
        bases = [(int(g) >> (2 * (19 - i))) & 3 for i in range(20)]
        for _ in range(copies):
            kind, pos = int(rng.integers(0, 3)), int(rng.integers(1, 19))
            tb = list(bases) if kind == 0 else (bases[:pos] + bases[pos + 1:] + [int(rng.integers(0, 4))] if kind == 1 else (bases[:pos] + [int(rng.integers(0, 4))] + bases[pos:])[:20])
            for _ in range(int(rng.integers(0, 4))):
                tb[int(rng.integers(0, 20))] = int(rng.integers(0, 4))
            v = 0
            for b in tb:
                v = (v << 2) | b
            extra.append(v)
    raw = np.unique(np.concatenate([raw, np.array(extra, dtype=np.uint64)]))
    pam_n = rng.integers(0, 4, size=len(raw)).astype(np.uint64)
    seq = (np.uint64(0b111111) << np.uint64(42)) | (pam_n << np.uint64(40)) | raw
    binkey = (seq >> np.uint64(2 * (24 - 11))) & np.uint64(0x3FFF)
    seq = seq[np.lexsort((seq, binkey))]
    return seq | (np.uint64(1) << np.uint64(48)), rng.integers(0, 1 << 27, size=len(seq), dtype=np.uint64)


@pytest.mark.parametrize("n_random,n_guides,max_mm,tttv", [(3_000_000, 1500, 3, True), (400_000, 3000, 2, False), (2_000, 40, 4, False)])
def test_cas12a_seeded_bulge_search_equals_brute_force_at_scale(capi, n_random, n_guides, max_mm, tttv):
    """the candidate search (prefix / shifted-suffix bucket seeds, csrc/ffh_bulge.hpp) must return exactly what the scan of every
    (guide, target) pair returns: same hits, same best alignment; bucket widths 8 .. 10 over the three database sizes"""
    rng = np.random.default_rng(n_random + max_mm)
    guides40 = rng.integers(0, 1 << 40, size=n_guides, dtype=np.uint64)
    targets, positions = _cas12a_database(rng, n_random, guides40[: min(n_guides, 400)], 12)
    guides = guides40 | np.uint64(0b11111100 << 40) | np.uint64(1 << 48)
    with capi.Context(1) as ctx:
        ctx.load_soa(targets, positions)
        widths = (ctx.info().prefix_bases, ctx.info().suffix_bases)
        res = ctx.discover_bulge(guides, max_mm, 1, tttv=tttv)
        bf = ctx.discover_bulge(guides, max_mm, 1, tttv=tttv, brute_force=True)
    assert widths[0] + widths[1] == 20
    for name in ("guide_offsets", "hit_targets", "hit_mismatches", "hit_bulge_type", "hit_bulge_position"):
        assert np.array_equal(getattr(res, name), getattr(bf, name)), name
    assert len(res.hit_targets) >= 400 and {0, 1, 2} <= set(res.hit_bulge_type.tolist())


@pytest.mark.parametrize("enzyme,prefix", [(5, 12), (5, 7), (6, 12), (6, 9)])
def test_19mer_packs_with_forced_splits_down_to_a_7_base_rest_key(capi, oracle, enzyme, prefix):
    """a 19-base pack with a 12-base bucket key leaves a 7-base rest key on the other image (group_words(7) = 16 words per
    group); the compare kernel has a row form for every rest width 7 .. 12 and the host refuses an image outside that range.  A
    database of clean 22-base sites in sequence order also makes the prefix image a DIRECT one (no slot -> index array)."""
    from tests.helpers import make_enzyme_case
    odb, t, p, g = make_enzyme_case(oracle, enzyme, 120000, 200, seed=3)
    with capi.Context(enzyme) as ctx:
        ctx.set_plan(prefix, 1)
        ctx.load_soa(t, p)
        info = ctx.info()
        assert (info.prefix_bases, info.suffix_bases) == (prefix, 19 - prefix)
        gpu = ctx.discover(g, 3, 2000, jost=True)
        gpu4 = ctx.discover(g[:50], 4, 40)
    ora = odb.discover(g, 3, 2000)
    assert_same_hits(gpu, ora)
    assert gpu.n_hits >= 50
    assert_same_scores(oracle, enzyme, g, gpu, ora, jost=True)
    assert_same_hits(gpu4, odb.discover(g[:50], 4, 40))


def test_direct_prefix_image_equals_the_indexed_one(capi, oracle, monkeypatch):
    """a 3'-PAM database in sequence order keeps its prefix buckets in database order and drops the slot -> index array
    (k_bucket_first); FFH_NO_DIRECT=1 builds the round-2 image with the array.  Same hits either way, and a database that is NOT in
    sequence order (ffh_db_load_soa takes what it is given) silently gets the indexed image."""
    odb, t, p, g = make_case(oracle, 100000, 300, enzyme=3, seed=9)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        a = ctx.discover(g, 4, 60, jost=True)
    assert_same_hits(a, odb.discover(g, 4, 60))
    monkeypatch.setenv("FFH_NO_DIRECT", "1")      # (read once per process: a no-op if another test built an image before; kept for A/B runs)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        b = ctx.discover(g, 4, 60, jost=True)
    monkeypatch.delenv("FFH_NO_DIRECT")
    assert a.summaries.tobytes() == b.summaries.tobytes() and np.array_equal(a.hit_targets, b.hit_targets)
    perm = np.random.default_rng(1).permutation(len(t))
    cnt = (t >> np.uint64(48)).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(cnt)])
    p2 = np.concatenate([p[off[i]:off[i + 1]] for i in perm])
    with capi.Context(3) as ctx:                                  # shuffled database order: hits per guide in THAT order
        ctx.load_soa(t[perm], p2)
        sh = ctx.discover(g, 4, 2 ** 31 - 1)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        full = ctx.discover(g, 4, 2 ** 31 - 1)
    order = {int(v): i for i, v in enumerate(t[perm])}
    for k in range(len(g)):
        assert sorted(sh.hits(k).tolist()) == sorted(full.hits(k).tolist()), k
        idx = [order[int(v)] for v in sh.hits(k)]
        assert idx == sorted(idx)


def test_error_paths_added_since_round_1(capi, oracle, monkeypatch):
    """(1) a guide set that collects 2^32 raw hits or more is not refused (round 5): ffh_discover halves it and concatenates the
    parts' results -- here every (guide, target) pair is a hit, 4.5e9 of them; the two-step ffh_scan still reports the limit;
    (2) a result whose page-locked block would exceed FFH_PINNED_LIMIT_MB fails with FFH_E_NOMEM and leaves the context usable;
    (3) image widths the compare kernel has no row form for are refused when the database is made resident"""
    rng = np.random.default_rng(77)
    raw = np.unique(rng.integers(0, 1 << 40, size=4_400_000, dtype=np.uint64))
    t = (raw << np.uint64(6)) | np.uint64(0b101010) | (np.uint64(1) << np.uint64(48))
    p = np.arange(len(t), dtype=np.uint64)
    g = (rng.integers(0, 1 << 40, size=1024, dtype=np.uint64) << np.uint64(6)) | np.uint64(0b101010) | (np.uint64(1) << np.uint64(48))
    assert len(t) * len(g) >= 2 ** 32
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        ctx.set_bounding(0)                                           # (bounding alone would keep this scan small: the split is what is tested)
        with pytest.raises(capi.FlashFryHipError, match="2\\^32 raw hits") as e:
            ctx.scan(g, 20)                                           # maxMismatch >= the guide length: every pair is a hit
        assert e.value.code == -1
        big = ctx.discover(g, 20, 2000, summaries_only=True)          # split in two: 2.25e9 raw hits each
        assert np.all(big.summaries["n_hits"] == 2000) and np.all(big.summaries["ot_count"] == 2000) and np.all(big.summaries["overflow"] == 1)
        assert big.n_hits == 2000 * len(g)
        ok = ctx.discover(g[:16], 3, 2000)                            # the context goes on
    monkeypatch.setenv("FFH_PINNED_LIMIT_MB", "1")                    # (read when a context is created: ffh_debug.hpp)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        with pytest.raises(capi.FlashFryHipError, match="pinned") as e:
            ctx.discover(g[:256], 9, 2 ** 31 - 1)                     # ~4e6 hits: a result block of tens of MB
        assert e.value.code == -6
        again = ctx.discover(g[:16], 3, 2000)
        assert again.summaries.tobytes() == ok.summaries.tobytes()
    monkeypatch.delenv("FFH_PINNED_LIMIT_MB")
    odb, t2, p2, g2 = make_case(oracle, 5000, 10, enzyme=3, seed=1)
    with capi.Context(3) as ctx:
        with pytest.raises(capi.FlashFryHipError, match="prefix_bases"):
            ctx.set_plan(13, 1)
        ctx.set_plan(6, 1)                                             # clamped to the supported range when the images are built
        ctx.load_soa(t2, p2)
        assert ctx.info().prefix_bases == 8 and ctx.info().suffix_bases == 12
        assert_same_hits(ctx.discover(g2, 4, 2000), odb.discover(g2, 4, 2000))


def test_work_list_that_outgrows_its_first_allocation_and_piled_up_candidates(capi, oracle, monkeypatch):
    """(1) the compare launch's work list starts at a size that suits a guide set without pile-ups and is cut off there; the host
    notices the cut after the launch, grows the list and runs the batch again (FFH_WORK_LIST_LIMIT forces a tiny first list);
    (2) many guides of one family on one huge bucket: the bucket's entries are split by candidate chunks as well as by groups
    (work_split) and every chunk of every piece reports its hits exactly once"""
    odb, t, p, g = dense_case(oracle, n_random=90_000, n_guides=700, n_dense=40, variants=150, seed=31)
    want = odb.discover(g, 4, 300)
    monkeypatch.setenv("FFH_WORK_LIST_LIMIT", "64")
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        for bounding in (0, 1):
            ctx.set_bounding(bounding)
            got = ctx.discover(g, 4, 300)
            assert_same_hits(got, want)
            assert_same_scores(oracle, 3, g, got, want)
    monkeypatch.delenv("FFH_WORK_LIST_LIMIT")
    # one family: 3000 near-copies of one 20-mer among random targets, 600 guides that are near-copies as well -> one prefix bucket
    # and one suffix bucket with hundreds of candidates and ~100 groups
    rng = np.random.default_rng(5)
    base = int(rng.integers(0, 1 << 40))
    def near(n, k):
        out = np.full(n, base, dtype=np.uint64)
        for i in range(n):
            for _ in range(int(rng.integers(0, k + 1))):
                out[i] ^= np.uint64(int(rng.integers(1, 4)) << (2 * int(rng.integers(4, 16))))   # (bases 2..7 and 12..19 stay: shared buckets)
        return out
    fam = np.unique(near(6000, 3))
    rest = rng.integers(0, 1 << 40, size=60_000, dtype=np.uint64)
    raw = np.unique(np.concatenate([fam, rest]))
    counts = rng.integers(1, 3, size=len(raw)).astype(np.uint64)
    t = (raw << np.uint64(6)) | np.uint64(0b101010) | (counts << np.uint64(48))
    n_pos = int(counts.sum())
    p = rng.integers(0, 1 << 27, size=n_pos, dtype=np.uint64) | (np.uint64(23) << np.uint64(52)) | (np.uint64(1) << np.uint64(32))
    g = (np.unique(near(900, 2)) << np.uint64(6)) | np.uint64(0b101010) | (np.uint64(1) << np.uint64(48))
    odb = oracle.db_from_sorted(3, t, p, contigs=["c1"])
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        for max_ot in (2 ** 31 - 1, 500):
            got, want = ctx.discover(g, 4, max_ot), odb.discover(g, 4, max_ot)
            assert_same_hits(got, want)
            assert_same_scores(oracle, 3, g, got, want)
        assert int(got.summaries["overflow"].sum()) > 0


@pytest.mark.parametrize("chunk", ["1", "4"])
def test_work_queues_of_the_compare_launch_on_short_lists(capi, oracle, monkeypatch, chunk):
    """the queue instances of k_compare are picked for long work lists (chunks of 16: hg38 scale -- the full-scale tests and the bench's
    own verification run it) and medium ones (chunks of 4: the slabs of a bounded scan at genome scale); FFH_WORK_QUEUE=1 / 4 forces
    them here, where most waves find their queue empty after their first chunk and the lists end inside a chunk -- every entry must
    still be taken exactly once"""
    monkeypatch.setenv("FFH_WORK_QUEUE", chunk)
    for seed, n_t, n_g, mm in ((3, 250_000, 400, 4), (4, 70_000, 150, 3), (5, 300, 7, 5)):
        odb, t, p, g = make_case(oracle, n_t, n_g, enzyme=3, seed=seed)
        with capi.Context(3) as ctx:
            ctx.load_soa(t, p)
            for bounding in (0, 1):
                ctx.set_bounding(bounding)
                got, want = ctx.discover(g, mm, 2000), odb.discover(g, mm, 2000)
                assert_same_hits(got, want)
    odb, t, p, g = dense_case(oracle, n_random=90_000, n_guides=500, n_dense=40, variants=150, seed=41)
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        got, want = ctx.discover(g, 4, 300), odb.discover(g, 4, 300)
        assert_same_hits(got, want)
        assert_same_scores(oracle, 3, g, got, want)


def test_repeated_scans_replay_the_captured_launch_sequence(capi, oracle):
    """From the third scan of a kind on, a context replays the candidate-list / work-list launches as one captured graph (PrepGraph,
    ffh_api.hip) -- while the guide COUNT, the plan, the images and the device buffers are unchanged.  The guides' CONTENT may change
    (the graph holds pointers, not values), a different count or mismatch budget must fall back to plain launches and later be
    captured on its own: every one of these calls against the oracle, and the first three answers again at the end."""
    odb, t, p, g = make_case(oracle, 180_000, 300, enzyme=3, seed=91)
    _, _, _, g2 = make_case(oracle, 1_000, 300, enzyme=3, seed=92)          # other guides, same count
    g2 = np.concatenate([g[:40], g2[40:]])                                    # (some with hits in this database)
    want = {}

    def check(ctx, guides, mm, tag):
        got = ctx.discover(guides, mm, 2000)
        if tag not in want:
            want[tag] = odb.discover(guides, mm, 2000)
        assert_same_hits(got, want[tag])
        return got

    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        for _ in range(4):                       # plain, plain (seen), captured, replayed
            check(ctx, g, 4, "a")
        for _ in range(2):
            check(ctx, g2, 4, "b")               # replayed with other guides in the same buffer
        check(ctx, g[:123], 4, "c")              # another count: plain launches
        for _ in range(4):
            check(ctx, g, 3, "d")                # another mismatch budget: its own capture
        got = check(ctx, g, 4, "a")
        again = check(ctx, g2, 4, "b")
        assert got.n_hits > 0 and again.n_hits > 0


def test_captured_sequence_is_not_replayed_over_another_budgets_patterns(capi, oracle):
    """the captured launches read the context's pattern lists, which a scan with another mismatch budget overwrites in
    place (smaller list, same allocation).  A, A, A, A (captured, replayed), B ONCE, A again must not replay A's graph over B's
    patterns; the same with B as a bounded scan and as a batched scan (neither is ever captured itself)."""
    odb, t, p, g = make_case(oracle, 180_000, 300, enzyme=3, seed=93)
    want = {mm: odb.discover(g, mm, 2000) for mm in (2, 3, 4)}
    with capi.Context(3) as ctx:
        ctx.load_soa(t, p)
        for _ in range(4):
            assert_same_hits(ctx.discover(g, 4, 2000), want[4])
        assert_same_hits(ctx.discover(g, 3, 2000), want[3])          # once: uploads the <= 3 patterns, captures nothing
        for _ in range(4):
            assert_same_hits(ctx.discover(g, 4, 2000), want[4])
        ctx.set_bounding(1)
        assert_same_hits(ctx.discover(g, 2, 2000), want[2])          # a bounded scan with a third budget
        ctx.set_bounding(0)
        for _ in range(4):
            assert_same_hits(ctx.discover(g, 4, 2000), want[4])
        os.environ["FFH_MAX_GUIDE_BATCH"] = "64"
        try:
            with capi.Context(3) as other:                               # (the batch limit is read when a context is created)
                other.load_soa(t, p)
                for _ in range(4):
                    assert_same_hits(other.discover(g[:64], 4, 2000), odb.discover(g[:64], 4, 2000))
                assert_same_hits(other.discover(g, 3, 2000), want[3])  # batched: five launches, never captured
                for _ in range(3):
                    assert_same_hits(other.discover(g[:64], 4, 2000), odb.discover(g[:64], 4, 2000))
        finally:
            del os.environ["FFH_MAX_GUIDE_BATCH"]
        assert_same_hits(ctx.discover(g, 4, 2000), want[4])


def test_a_guide_set_with_more_raw_hits_than_one_scan_holds_is_split_not_refused(capi, oracle, monkeypatch):
    """where one scan would collect more raw hits than its 32-bit segment arithmetic holds (FFH_RAW_HIT_LIMIT puts
    that limit at 2048 here) the library first bounds the scan and then halves the guide set, as often as needed, and concatenates the
    parts: lists, positions, per-hit scores and aggregates must be those of the oracle's single pass -- <= 6 mismatches on a database
    with dense neighbourhoods, bounding on, off and automatic, lists and aggregates-only."""
    odb, t, p, g = dense_case(oracle, n_random=120_000, n_guides=500, n_dense=60, variants=200, seed=57)
    monkeypatch.setenv("FFH_RAW_HIT_LIMIT", "2048")   # (the scans below collect 5 000 - 12 000 raw hits: several levels of halving)
    for mm, max_ot in ((6, 2000), (5, 40), (6, 2 ** 31 - 1)):
        want = odb.discover(g, mm, max_ot)
        for bounding in (0, 1, -1):
            with capi.Context(3) as ctx:
                ctx.load_soa(t, p)
                ctx.set_bounding(bounding)
                with pytest.raises(capi.FlashFryHipError, match="2\\^32 raw hits"):
                    ctx.scan(g, mm)                                   # the two-step entry point reports the limit
                got = ctx.discover(g, mm, max_ot, jost=True)
                assert_same_hits(got, want)
                if mm == 5:
                    assert_same_scores(oracle, 3, g, got, want, jost=True)
                only = ctx.discover(g, mm, max_ot, summaries_only=True, jost=True)
                assert only.summaries.tobytes() == got.summaries.tobytes() and only.n_hits == got.n_hits
                slim = ctx.discover(g, mm, max_ot, positions=False, hit_scores=False)
                assert np.array_equal(slim.hit_targets, got.hit_targets) and np.array_equal(slim.guide_offsets, got.guide_offsets)
    monkeypatch.delenv("FFH_RAW_HIT_LIMIT")


def test_bulge_search_splits_a_guide_set_with_more_candidate_records_than_one_search_holds(capi, monkeypatch):
    """the bulge path of VERDICT r4 next 7: with the record limit at 512 (FFH_RAW_HIT_LIMIT) the seeded search and the brute force halve
    the guide set as often as needed; the concatenated result is the unsplit one"""
    rng = np.random.default_rng(4242)
    guides40 = rng.integers(0, 1 << 40, size=300, dtype=np.uint64)
    targets, positions = _cas12a_database(rng, 60_000, guides40, 12)
    guides = guides40 | np.uint64(0b11111100 << 40) | np.uint64(1 << 48)
    with capi.Context(1) as ctx:
        ctx.load_soa(targets, positions)
        whole = ctx.discover_bulge(guides, 3, 1)
    assert len(whole.hit_targets) > 2000
    monkeypatch.setenv("FFH_RAW_HIT_LIMIT", "512")
    with capi.Context(1) as ctx:
        ctx.load_soa(targets, positions)
        parts = ctx.discover_bulge(guides, 3, 1)
        brute = ctx.discover_bulge(guides[:64], 3, 1, brute_force=True)
    monkeypatch.delenv("FFH_RAW_HIT_LIMIT")
    for name in ("guide_offsets", "hit_targets", "hit_mismatches", "hit_bulge_type", "hit_bulge_position"):
        assert np.array_equal(getattr(parts, name), getattr(whole, name)), name
    n64 = int(whole.guide_offsets[64])
    assert np.array_equal(brute.hit_targets, whole.hit_targets[:n64]) and np.array_equal(brute.guide_offsets, whole.guide_offsets[:65])
