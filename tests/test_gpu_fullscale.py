"""The hot path at BASELINE.json's full size -- config C3: 100 000 guides against 3.0e8 targets (hg38 scale), <= 4 mismatches --
checked through size-independent properties:
an independent brute-force torch scan of all targets for a sample of guides, database order, the cut-off rule, planted
copies at every mismatch level, invariance under the candidate split and under bin sharding.  No oracle here: it would
need hours at this size (it is the checker at the small sizes in test_gpu_parity.py)."""
import hashlib

import numpy as np
import pytest

from flashfry_amd import synth

pytestmark = pytest.mark.gpu

T_FULL = int(3.0e8)
G = 100000                       # BASELINE.json configs[2]
MAX_OT = 2000
CMP_MASK = 0x3FFFFFFFFFC0          # StandardScanParameters.scala:143
UPPER = 0xAAAAAAAAAAAA             # BitEncoding.scala:205


def digest(res):
    h = hashlib.sha256()
    for a in (res.guide_offsets, res.hit_targets, res.hit_mismatches, res.pos_offsets, res.positions, res.summaries):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def torch_mismatches(torch, guide, targets):
    """BitEncoding.mismatches :127-132 written with torch integer ops over the whole database"""
    x = (targets ^ guide) & CMP_MASK
    y = (x | (x << 1)) & UPPER
    y = y - ((y >> 1) & 0x5555555555555555)
    y = (y & 0x3333333333333333) + ((y >> 2) & 0x3333333333333333)
    y = (y + (y >> 4)) & 0x0F0F0F0F0F0F0F0F
    return (y * 0x0101010101010101) >> 56 & 0x7F


@pytest.fixture(scope="module")
def world():
    import torch
    from flashfry_amd import capi
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    guides_dev = synth.make_guides(G, device=dev)
    db = synth.make_database(T_FULL, seed=synth.DB_SEED, plant_guides=guides_dev, device=dev)
    ctx = capi.Context(3)
    torch.cuda.synchronize()
    ctx.load_soa_device(db["targets"].data_ptr(), db["T"], db["positions"].data_ptr(), db["P"])
    guides = guides_dev.cpu().numpy().view(np.uint64)
    yield dict(torch=torch, capi=capi, ctx=ctx, db=db, guides=guides, guides_dev=guides_dev)
    ctx.close()


def test_every_hit_of_sampled_guides_matches_a_brute_force_torch_scan(world):
    torch, ctx, db = world["torch"], world["ctx"], world["db"]
    # 40 guides: planted ones (every 100th guide has copies at 0..4 mismatches in the database) and plain ones, spread over the
    # whole guide range so that every part of the candidate lists / every guide batch is sampled
    sample = list(range(0, G, 100))[::50] + [1, 2, 3, 777, 4999, 50001, 65537, 99998, 99999] + list(range(12345, G, 9973))
    assert len(set(sample)) >= 32
    res = ctx.discover(world["guides"], 4, 2 ** 31 - 1)         # no cut-off: the complete hit sets
    for g in sample:
        mm = torch_mismatches(torch, int(world["guides"][g].astype(np.int64)), db["targets"])
        idx = torch.nonzero(mm <= 4).flatten()
        want = db["targets"][idx].cpu().numpy().view(np.uint64)   # database order = ascending index
        got = res.hits(g)
        assert np.array_equal(got, want), g
        a, b = int(res.guide_offsets[g]), int(res.guide_offsets[g + 1])
        assert np.array_equal(res.hit_mismatches[a:b], mm[idx].cpu().numpy().astype(np.uint8))
        po = res.pos_offsets[a:b + 1]
        src = db["pos_offsets"][idx].cpu().numpy()
        for k in (0, len(idx) // 2, len(idx) - 1):                # positions of the first, middle and last hit
            if len(idx):
                n = int(po[k + 1] - po[k])
                assert n == int(want[k] >> np.uint64(48))
                assert np.array_equal(res.positions[int(po[k]):int(po[k + 1])], db["positions"][int(src[k]):int(src[k]) + n].cpu().numpy().view(np.uint64))


def test_aggregates_of_sampled_guides_are_exact_at_full_size(world, oracle):
    """CFD / Hsu2013 / CRISPRi / closest-hit aggregates at config C3's size, EXACT for 240 sampled guides (they were only
    range-checked here): the brute-force torch scan gives a guide's complete hit list in database order, the ordered cut-off of
    CRISPRSiteOT.addOT / full (crispr/CRISPRSiteOT.scala:39-46) is applied to it in numpy, and the oracle's string-level score_guide
    (Doench2016CFDScore.scala:53-88, CrisprMitEduOffTarget.scala:98-148, ClosestHit.scala:57-67) folds the retained list; the library's
    per-guide summaries -- from the aggregates-only step bench.py times -- must equal that bit for bit, with the cut-off far away
    (2000) and biting (60)."""
    from tests.helpers import assert_same_scores
    torch, ctx, db = world["torch"], world["ctx"], world["db"]
    sample = sorted(set(list(range(0, G, 100))[::8] + list(range(7, G, 863))))[:240]
    assert len(sample) >= 200
    lists = {}
    for g in sample:
        mm = torch_mismatches(torch, int(world["guides"][g].astype(np.int64)), db["targets"])
        idx = torch.nonzero(mm <= 4).flatten()
        lists[g] = db["targets"][idx].cpu().numpy().view(np.uint64)
    sub = world["guides"][sample]
    for max_ot in (MAX_OT, 60):
        ctx.scan(world["guides"], 4)
        only = ctx.finalize(max_ot, summaries_only=True, jost=True)          # the step the bench times: all 100 000 guides, aggregates only
        full = ctx.discover(sub, 4, max_ot, jost=True)                       # the sampled guides with their lists and per-hit scores
        assert only.summaries[sample].tobytes() == full.summaries.tobytes()

        class Ora:   # what assert_same_scores reads of an oracle result
            n_guides = len(sample)

            def hits(self, k):
                h = lists[sample[k]]
                before = np.concatenate([[0], np.cumsum((h >> np.uint64(48)).astype(np.int64))[:-1]])
                return h[before < max_ot]

        ora = Ora()
        for k in range(len(sample)):
            assert np.array_equal(full.hits(k), ora.hits(k)), (max_ot, sample[k])
        assert_same_scores(oracle, 3, sub, full, ora, exact=True, jost=True)
        if max_ot == 60:
            assert int(full.summaries["overflow"].sum()) >= 20               # (the cut-off did bite)


def test_order_cutoff_and_planted_copies(world):
    ctx = world["ctx"]
    res = ctx.discover(world["guides"], 4, MAX_OT, jost=True)
    s = res.summaries
    seq = res.hit_targets & np.uint64((1 << 48) - 1)
    cnt = (res.hit_targets >> np.uint64(48)).astype(np.int64)
    off = res.guide_offsets.astype(np.int64)
    gid = np.repeat(np.arange(G), np.diff(off))
    assert np.all((seq[1:] > seq[:-1]) | (gid[1:] != gid[:-1]))   # strictly ascending sequences inside a guide = database order, no duplicates
    assert res.hit_mismatches.max() <= 4
    csum = np.concatenate([[0], np.cumsum(cnt)])
    tot = csum[off[1:]] - csum[off[:-1]]
    assert np.array_equal(tot, s["ot_count"].astype(np.int64))   # otCount = positions of the retained hits
    assert np.array_equal(s["overflow"].astype(bool), tot >= MAX_OT)                  # CRISPRSiteOT.full
    last = np.where(off[1:] > off[:-1], cnt[np.maximum(off[1:] - 1, 0)], 0)
    assert np.all((tot - last)[s["overflow"].astype(bool)] < MAX_OT)                  # the hit that crossed the limit was the last one added
    planted = s[::100]
    ok = ~planted["overflow"].astype(bool)
    assert ok.sum() >= 30 and np.all(planted["hist"][ok] >= 1)    # an exact copy and copies at 1..4 mismatches were planted
    assert np.all(planted["in_genome"][ok] >= 1) and np.all(planted["closest"][ok] == 1)
    scored = s["n_scored"] > 0
    assert np.all((s["cfd_max"][scored] > 0) & (s["cfd_max"][scored] <= 1.0) & (s["cfd_sum"][scored] >= s["cfd_max"][scored]))
    assert np.all((s["jost_max"][scored] > 0) & (s["jost_sum"][scored] >= s["jost_max"][scored] - 1e-12))
    world["reference_digest"] = digest(res)


def test_result_does_not_depend_on_the_candidate_split_or_on_sharding(world):
    torch, capi, ctx, db = world["torch"], world["capi"], world["ctx"], world["db"]
    if "reference_digest" not in world:
        world["reference_digest"] = digest(ctx.discover(world["guides"], 4, MAX_OT, jost=True))
    ctx.set_plan(10, 1)
    try:
        assert digest(ctx.discover(world["guides"], 4, MAX_OT, jost=True)) == world["reference_digest"]
        tm = ctx.timings()
        assert (tm.prefix_bases, tm.prefix_radius) == (10, 1)
    finally:
        ctx.set_plan(-1, -1)
    # two shards at a target boundary, ordered cut-off continued across them
    whole = ctx.discover(world["guides"], 4, MAX_OT)
    cut = db["T"] // 2 + 12345
    pcut = int(db["pos_offsets"][cut])
    with capi.Context(3) as c0, capi.Context(3) as c1:
        c0.load_soa_device(db["targets"].data_ptr(), cut, db["positions"].data_ptr(), pcut)
        c1.load_soa_device(db["targets"][cut:].data_ptr(), db["T"] - cut, db["positions"][pcut:].data_ptr(), db["P"] - pcut)
        c0.scan(world["guides"], 4)
        c1.scan(world["guides"], 4)
        r0 = c0.finalize(MAX_OT)
        r1 = c1.finalize(MAX_OT, prior_totals=c0.shard_totals(MAX_OT))
    n0, n1 = np.diff(r0.guide_offsets.astype(np.int64)), np.diff(r1.guide_offsets.astype(np.int64))
    assert np.array_equal(n0 + n1, np.diff(whole.guide_offsets.astype(np.int64)))
    merged = np.concatenate([np.concatenate([r0.hits(g), r1.hits(g)]) for g in range(0, G, 97)])
    assert np.array_equal(merged, np.concatenate([whole.hits(g) for g in range(0, G, 97)]))
    assert np.array_equal(r0.summaries["ot_count"] + r1.summaries["ot_count"], whole.summaries["ot_count"])
    assert np.array_equal(r0.summaries["overflow"] | r1.summaries["overflow"], whole.summaries["overflow"])


def test_five_and_three_mismatches_take_the_10_10_images_and_four_the_11_9_ones(world):
    """select_images: at hg38 scale the cost model prefers a 10 + 10 split for <= 5 (and <= 3) mismatches and 11 + 9 for <= 4; the context
    builds the second pair of images on first use and swaps per call.  Complete hit sets of sampled guides against the brute-force
    torch scan under both pairs."""
    torch, ctx, db = world["torch"], world["ctx"], world["db"]
    g = world["guides"][:3000]
    ctx.set_plan(-1, -1)
    for max_mm, width in ((5, 10), (4, 11), (3, 10), (4, 11)):
        res = ctx.discover(g, max_mm, 2 ** 31 - 1, positions=False, hit_scores=False)
        assert ctx.info().prefix_bases == width and ctx.timings().prefix_bases == width, (max_mm, ctx.info().prefix_bases)
        for k in (0, 1, 100, 700, 1500, 2999):
            mm = torch_mismatches(torch, int(g[k].astype(np.int64)), db["targets"])
            idx = torch.nonzero(mm <= max_mm).flatten()
            assert np.array_equal(res.hits(k), db["targets"][idx].cpu().numpy().view(np.uint64)), (max_mm, k)


def test_c4_eight_bin_shards_through_the_library_exchange_equal_the_unsharded_discover(world):
    """Config C4 of BASELINE.json at its stated shape on the one GPU of the test box: the SAME 3.0e8-target database split into EIGHT
    contiguous bin shards balanced by payload (dist.shard_bins, the role of BinaryHeader.uncompressedSize, BinaryHeader.scala:54),
    eight contexts, ONE ffh_discover_sharded (scan of every shard + totals all-gather + prior + fix-up + the three reductions; the
    copy transport, since RCCL refuses eight ranks on one device).  maximumOffTargets is set so low that for most guides the
    ordered cut-off (CRISPRSiteOT.scala:39-46) is reached only after the running total has crossed several shard boundaries.  The
    reduced aggregates and the concatenated per-shard hit lists must be the unsharded discover's."""
    torch, capi, ctx, db = world["torch"], world["capi"], world["ctx"], world["db"]
    from flashfry_amd import dist as ffdist
    guides = world["guides"]
    W, max_ot = 8, 60
    T = db["T"]
    bins = (db["targets"] >> 32) & 0x3FFF                               # 7-base bin = bits 45:32 of a Cas9 23-mer
    per_bin_t = torch.bincount(bins, minlength=16384)
    first = torch.cumsum(per_bin_t, 0) - per_bin_t
    bin_pos = db["pos_offsets"][torch.clamp(first + per_bin_t, max=T)] - db["pos_offsets"][torch.clamp(first, max=T)]
    payload = ((per_bin_t + bin_pos) * 8).cpu().numpy()
    cuts = ffdist.shard_bins(payload, W)
    assert all(b1 > b0 for b0, b1 in cuts)
    whole = ctx.discover(guides, 4, max_ot, jost=True, positions=False, hit_scores=False)
    ctxs = []
    try:
        for b0, b1 in cuts:
            lo = int(first[b0]) if b0 < 16384 else T
            hi = int(first[b1]) if b1 < 16384 else T
            plo, phi = int(db["pos_offsets"][lo]), int(db["pos_offsets"][hi])
            c = capi.Context(3)
            c.load_soa_device(db["targets"][lo:hi].data_ptr(), hi - lo, db["positions"][plo:phi].data_ptr(), phi - plo)
            ctxs.append(c)
        assert sum(c.info().n_targets for c in ctxs) == T
        with capi.Comm.local(ctxs) as comm:
            assert comm.world == W and comm.transport == "copy"
            summ = comm.discover(guides, 4, max_ot, jost=True)
            lists = [comm.shard_lists(i, positions=False, hit_scores=False) for i in range(W)]
    finally:
        for c in ctxs:
            c.close()
    s = whole.summaries
    ints = ("n_hits", "ot_count", "overflow", "hist", "closest", "closest_count", "in_genome", "n_scored")
    h0, h1 = hashlib.sha256(), hashlib.sha256()
    for f in ints + ("cfd_max", "jost_max"):
        assert np.array_equal(s[f], summ[f]), f
        h0.update(np.ascontiguousarray(s[f]).tobytes()); h1.update(np.ascontiguousarray(summ[f]).tobytes())
    assert h0.hexdigest() == h1.hexdigest()
    for f in ("cfd_sum", "hsu_sum", "jost_sum"):      # added shard by shard: equal up to the association of the additions
        assert np.abs(s[f] - summ[f]).max() <= 1e-9 * max(1.0, float(np.abs(s[f]).max())), f
    counts = np.stack([np.diff(l.guide_offsets.astype(np.int64)) for l in lists])       # [shard][guide] retained hits
    assert np.array_equal(counts.sum(0), np.diff(whole.guide_offsets.astype(np.int64)))
    for g in range(0, G, 97):
        assert np.array_equal(np.concatenate([l.hits(g) for l in lists]), whole.hits(g)), g
    # the cut-off really was decided across shards: overflowed guides whose retained hits span at least four shards (three boundaries)
    over = s["overflow"].astype(bool)
    spans = (counts > 0).sum(0)
    assert over.sum() > G // 2 and int((over & (spans >= 4)).sum()) > 1000, (int(over.sum()), int((over & (spans >= 4)).sum()))
    assert np.array_equal(summ["ot_count"] >= max_ot, over)                     # CRISPRSiteOT.full on the reduced totals
