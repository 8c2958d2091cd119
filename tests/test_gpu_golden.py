"""The reference's own golden fixtures and known answers put through the PRODUCT -- the HIP library behind the C ABI -- not through the
oracle: nothing in this file loads `oracle/`.  What is compared with what:

  tests/golden/test_blockAACCTTGG.binary   (BlockManagerTest.scala:105-131: the same targets as a linear and as an indexed block give the
                                           same hits)  ->  ffh_db_load_blocks / ffh_db_write + ffh_db_open + ffh_discover against a numpy
                                           brute force over the fixture's longs
  tests/golden/known_answers.json          literals of Doench2016CFDScoreTest, CrisprMitEduOffTargetTest, ClosestHitTest,
                                           JoistAndSantosCRISPRiTest -> ffh_score_lists; SimpleSiteFinderTest -> the device site scanner
                                           (ffh_indexer_*) + ffh_db_open + ffh_discover
  tests/golden/fake.sites                  the 9 255 (guide, off-target, mismatches) triples of the reference's table fixture ->
                                           ffh_score_lists' per-hit mismatch counts and per-guide totals
The strings are encoded here in plain Python (BitEncoding.scala:46-67: A C G T = 0 1 2 3, first base in the most significant position,
count << 48); the brute force is numpy on the raw longs (BitEncoding.scala:127-132)."""
import json
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCAN = {1: 24, 2: 23, 3: 23, 4: 23, 5: 22, 6: 22}
CMP_MASK = {1: 0x00FFFFFFFFFF, 2: 0x3FFFFFFFFFC0, 3: 0x3FFFFFFFFFC0, 4: 0x3FFFFFFFFFC0, 5: 0x0FFFFFFFFFC0, 6: 0x0FFFFFFFFFC0}   # StandardScanParameters.scala:99-205


def enc(s, count=1):
    v = 0
    for ch in s:
        v = (v << 2) | "ACGT".index(ch)
    return v | (count << 48)


def dec(v, n):
    return "".join("ACGT"[(int(v) >> (2 * (n - 1 - i))) & 3] for i in range(n))


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def mismatches_np(enzyme, guide, targets):
    """BitEncoding.mismatches on an array of target longs (bitcoding/BitEncoding.scala:127-132)"""
    x = (targets ^ np.uint64(guide)) & np.uint64(CMP_MASK[enzyme])
    f = (x | (x >> np.uint64(1))) & np.uint64(0x555555555555)
    f = f - ((f >> np.uint64(1)) & np.uint64(0x5555555555555555))
    f = (f & np.uint64(0x3333333333333333)) + ((f >> np.uint64(2)) & np.uint64(0x3333333333333333))
    f = (f + (f >> np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    return ((f * np.uint64(0x0101010101010101)) >> np.uint64(56)).astype(np.int64)


@pytest.fixture(scope="module")
def capi():
    from flashfry_amd import capi
    assert capi.load_library().ffh_device_count() >= 1, "no GPU visible: the gpu tests must run on the MI355X box"
    return capi


@pytest.fixture(scope="module")
def ka(golden_dir):
    with open(os.path.join(golden_dir, "known_answers.json")) as f:
        return json.load(f)


def block_fixture(golden_dir):
    raw = open(os.path.join(golden_dir, "test_blockAACCTTGG.binary"), "rb").read()
    vals = struct.unpack(">%dq" % (len(raw) // 8), raw)   # DataOutputStream: big-endian; [n][target, position] * (BlockManagerTest.scala:105-118)
    assert vals[0] == len(vals) - 1 == 20260
    return np.array(vals[1:], dtype=np.int64)


# ---- BlockManagerTest.scala:105-131 -------------------------------------------------------------------------------------------
def test_block_fixture_linear_and_indexed_through_the_library(capi, golden_dir, tmp_path):
    body = block_fixture(golden_dir)
    targets, positions = body[0::2].view(np.uint64), body[1::2].view(np.uint64)
    assert dec(targets[0], 23).startswith("AACCTTGG") and np.all(targets >> np.uint64(48) == 1)
    bin_idx = int(targets[0] >> np.uint64(32)) & 0x3FFF           # the 7-base bin AACCTTG
    # guides: members of the block with 0-3 substitutions behind the 8-base prefix, a few unrelated ones
    rng = np.random.default_rng(9)
    guides = []
    for k in range(400):
        s = list(dec(targets[rng.integers(len(targets))], 23))
        for p in rng.choice(np.arange(8, 20), size=rng.integers(0, 4), replace=False):
            s[p] = rng.choice(list("ACGT"))
        guides.append(enc("".join(s)))
    guides += [enc("".join(rng.choice(list("ACGT"), 20)) + "TGG") for _ in range(20)]
    guides = np.array(guides, dtype=np.uint64)

    def check(ctx, max_mm, max_ot):
        r = ctx.discover(guides, max_mm, max_ot)
        total = 0
        for g in range(len(guides)):
            mm = mismatches_np(2, int(guides[g]), targets)
            want = targets[mm <= max_mm][:max_ot]                  # count 1 each: the cut-off keeps the first max_ot hits in database order
            a, b = int(r.guide_offsets[g]), int(r.guide_offsets[g + 1])
            assert np.array_equal(r.hit_targets[a:b], want), g
            assert np.array_equal(r.hit_mismatches[a:b], mm[mm <= max_mm][:max_ot])
            idx = np.nonzero(mm <= max_mm)[0][:max_ot]
            assert np.array_equal(r.positions[int(r.pos_offsets[a]):int(r.pos_offsets[b])], positions[idx])   # one position per target
            assert int(r.summaries["overflow"][g]) == (len(want) >= max_ot) and int(r.summaries["ot_count"][g]) == len(want)
            total += len(want)
        assert r.n_hits == total
        return r

    # (1) the fixture as ONE LINEAR block (type long 1, BlockManager.scala:74-79) in its bin, every other bin an empty linear block
    longs = np.ones(16384 + len(body), dtype=np.int64)
    offs = np.arange(16385, dtype=np.uint64)
    offs[bin_idx + 1:] += np.uint64(len(body))
    longs[bin_idx + 1:bin_idx + 1 + len(body)] = body
    with capi.Context(2) as ctx:
        ctx.load_blocks(longs, offs)
        assert ctx.info().n_targets == 10130 and ctx.info().n_positions == 10130
        lin3 = check(ctx, 3, 100000)
        lin4 = check(ctx, 4, 37)
        assert lin3.n_hits >= 400
    # (2) the same targets through the product's database writer: 10 130 targets in one bin is an INDEXED block (type long 2 +
    #     256-entry lookup table, BlockManager.scala:362-442), read back by ffh_db_open
    path = tmp_path / "fixture_db"
    capi.write_database(path, 2, targets, positions, ["chr22"], bin_width=7)
    import gzip
    hdr = open(str(path) + ".header").read().split("\n")
    line = [ln for ln in hdr if ln.startswith("AACCTTG=")][0]
    off, nbytes, ntargets = (int(x) for x in line.split("=")[1].split(","))
    blk = np.frombuffer(gzip.open(str(path), "rb").read()[off:off + nbytes], dtype="<i8")   # (Utils.longArrayToByteArray: little-endian, UtilsTest.scala:38-46)
    assert ntargets == 10130 and blk[0] == 2 and len(blk) == 1 + 256 + len(body)
    with capi.Context(0) as ctx:
        ctx.open(str(path))
        idx3 = check(ctx, 3, 100000)
        idx4 = check(ctx, 4, 37)
    # "the same hits from both block types": the statement of the reference test
    for a, b in ((lin3, idx3), (lin4, idx4)):
        assert np.array_equal(a.guide_offsets, b.guide_offsets) and np.array_equal(a.hit_targets, b.hit_targets)
        assert a.summaries.tobytes() == b.summaries.tobytes()


# ---- Doench2016CFDScoreTest.scala / CrisprMitEduOffTargetTest.scala / ClosestHitTest.scala / JoistAndSantosCRISPRiTest.scala ----
def score(capi, enzyme, guide, hits):
    """one guide with a list of (sequence, count) off-targets through ffh_score_lists"""
    g = np.array([enc(guide)], dtype=np.uint64)
    t = np.array([enc(s, c) for s, c in hits], dtype=np.uint64)
    with capi.Context(enzyme) as ctx:
        return ctx.score_lists(g, np.array([0, len(t)], dtype=np.uint64), t)


def test_cfd_known_answers_through_score_lists(capi, ka):
    c = ka["cfd_pairs"]                                               # scoreCFD(guide 20-mer, off-target 20-mer): the PAM factor is 1.0 for GG
    r = score(capi, 2, c["guide"] + "TGG", [(ot + "TGG", 1) for ot, _ in c["cases"]])
    assert r.scores_valid
    for got, (_, exp) in zip(r.hit_cfd, c["cases"]):
        assert got == pytest.approx(exp, abs=c["tol"]), c["source"]
    for case in ka["cfd_guides"]:                                     # scoreGuide(...)(0)(0): the maximum, printed as 0.0 below 0.023 (:83-87)
        r = score(capi, 2, case["guide"], [(h, 1) for h in case["hits"]])
        m = float(r.summaries["cfd_max"][0])
        assert (m if m >= 0.023 else 0.0) == pytest.approx(case["maxOT_printed"], abs=case["tol"]), case["source"]
    # the list of the third case contains the guide itself: skipped (Doench2016CFDScore.scala:67), so one hit is unscored
    assert int(r.summaries["n_scored"][0]) == len(ka["cfd_guides"][2]["hits"]) - 1 and np.isnan(r.hit_cfd).sum() == 1


def test_hsu_known_answers_through_score_lists(capi, ka):
    c = ka["hsu_guide"]
    r = score(capi, 2, c["guide"], [(h, 1) for h in c["hits"]])
    assert float(r.hsu2013()[0]) == pytest.approx(c["expected"], abs=c["tol"]), c["source"]
    p = ka["hsu_pair"]                                                # one off-target: the guide's sum is that pair's score
    r = score(capi, 2, p["guide"], [(p["ot"], 1)])
    assert float(r.summaries["hsu_sum"][0]) == pytest.approx(p["expected"], abs=p["tol"]), p["source"]


def test_closest_hit_known_answers_through_score_lists(capi, ka):
    rng = np.random.default_rng(5)
    for c in ka["closest_cases"]:
        g, hits = c["guide"], []
        for mm, cnt in zip(c["mm"], c["counts"]):
            while True:
                s = list(g)
                for p in rng.choice(20, size=mm, replace=False):
                    s[p] = rng.choice([b for b in "ACGT" if b != g[p]])
                s = "".join(s)
                if s not in [h[0] for h in hits]:
                    break
            hits.append((s, cnt))
        m = score(capi, 2, g, hits).summaries[0]
        assert str(int(m["closest"])) == c["closest"] and str(int(m["closest_count"])) == c["count"], c["source"]
        assert ",".join(str(int(x)) for x in m["hist"]) == c["hist"], c["source"]
    m = score(capi, 2, g, []).summaries[0]
    assert int(m["closest"]) == 0xFFFFFFFF and int(m["closest_count"]) == 0          # printed "UNK", "0"


def test_jost_known_answers_through_score_lists(capi, ka):
    def product(f):
        x = 1.0
        for v in f:
            x *= v
        return x
    for enzyme, target, off, factors, src in ka["jost_pairs"]:
        m = score(capi, enzyme, target, [(off, 1)]).summaries[0]
        assert float(m["jost_max"]) == product(factors) and float(m["jost_sum"]) == product(factors), src
    for enzyme, guide, hits, factors, src in ka["jost_guides"]:
        m = score(capi, enzyme, guide, [(h, 1) for h in hits]).summaries[0]
        assert float(m["jost_max"]) == (product(factors) if factors else 0.0), src
        if factors:
            assert 1.0 / (1.0 + float(m["jost_sum"])) == 1.0 / (1.0 + product(factors)), src


# ---- test_data/fake.sites (TabDelimitedHanderTest.scala:40-51) ---------------------------------------------------------------------
def test_fake_sites_triples_through_score_lists(capi, golden_dir):
    guides, offs, hits, want_mm, want_tot = [], [0], [], [], []
    for ln in open(os.path.join(golden_dir, "fake.sites")).read().split("\n")[1:]:
        if not ln:
            continue
        f = ln.split("\t")
        guides.append(enc(f[3]))
        for tok in f[8].split(","):
            seq, cnt, rest = tok.split("_", 2)
            hits.append(enc(seq, int(cnt)))
            want_mm.append(int(rest.split("<")[0]))
        offs.append(len(hits))
        want_tot.append(int(f[7]))
    assert len(hits) == 9255
    with capi.Context(2) as ctx:
        r = ctx.score_lists(np.array(guides, dtype=np.uint64), np.array(offs, dtype=np.uint64), np.array(hits, dtype=np.uint64))
    assert np.array_equal(r.hit_mismatches, np.array(want_mm, dtype=np.uint8))          # the _N field of every off-target token
    assert np.array_equal(r.summaries["ot_count"], np.array(want_tot, dtype=np.uint32))  # the otCount column
    assert np.array_equal(r.summaries["n_hits"], np.diff(np.array(offs)).astype(np.uint32))
    hist = np.zeros((len(guides), 5), dtype=np.int64)
    for g in range(len(guides)):
        for h in range(offs[g], offs[g + 1]):
            if want_mm[h] < 5:
                hist[g, want_mm[h]] += int(hits[h]) >> 48
    assert np.array_equal(r.summaries["hist"].astype(np.int64), hist)


# ---- SimpleSiteFinderTest.scala:13-173 ---------------------------------------------------------------------------------------------
def test_site_finder_known_answers_through_the_device_indexer(capi, ka, tmp_path):
    for k, (enzyme, flank, seq, expected, src) in enumerate(ka["site_cases"]):
        path = tmp_path / ("sites%d" % k)
        st = capi.index_contigs(path, enzyme, [("ctg", seq)], bin_width=3)
        L = SCAN[enzyme]
        assert st.n_sites == len(expected), src
        uniq = sorted({e[0] for e in expected})
        assert st.n_targets == len(uniq), src
        guides = np.array([enc(s) for s in uniq], dtype=np.uint64)
        with capi.Context(0) as ctx:
            ctx.open(str(path))
            r = ctx.discover(guides, 0, 1000)
        for g, s in enumerate(uniq):
            a, b = int(r.guide_offsets[g]), int(r.guide_offsets[g + 1])
            assert b - a == 1 and dec(r.hit_targets[a], L) == s, src
            pos = r.positions[int(r.pos_offsets[a]):int(r.pos_offsets[b])]
            got = sorted((int(p) & 0xFFFFFFFF, (int(p) >> 60) == 0, (int(p) >> 52) & 0xFF, (int(p) >> 32) & 0xFFFFF) for p in pos)   # BitPosition.scala:51-72
            want = sorted((e[1], e[2], L, 1) for e in expected if e[0] == s)
            assert got == want, src
            for start, fwd, _, _ in got:
                assert s == (seq[start:start + L] if fwd else revcomp(seq[start:start + L])), src
