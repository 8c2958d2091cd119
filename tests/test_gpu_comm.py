"""The bin-sharded discover as one library call (ffh_comm_*, csrc/ffh_comm.hpp): scans + the exchange of SURVEY.md section 8e with the
collectives issued inside the library.  On the one GPU of the test box: several shards on one device (the copy transport -- RCCL
refuses duplicate devices) and a one-rank RCCL communicator (ncclCommInitRank, the collectives really go through librccl).  The
reference for every comparison is the unsharded discover of the same database, which test_gpu_parity.py checks against the oracle."""
import numpy as np
import pytest

from flashfry_amd import dist as ffdist

pytestmark = pytest.mark.gpu

INTS = ("n_hits", "ot_count", "overflow", "hist", "closest", "closest_count", "in_genome", "n_scored")


@pytest.fixture(scope="module")
def capi():
    from flashfry_amd import capi as c
    c.load_library()
    return c


@pytest.fixture(scope="module")
def case(oracle):
    from tests.test_gpu_parity import dense_case
    odb, targets, positions, guides = dense_case(oracle, n_random=200000, n_guides=300, n_dense=40, variants=120, seed=31)
    sizes = [len(odb.bin(b)[0]) * 8 for b in range(odb.n_bins)]
    return odb, targets, positions, guides, sizes


def shard_slices(targets, sizes, world):
    binidx = ((targets >> np.uint64(32)) & np.uint64(0x3FFF)).astype(np.int64)
    poff = np.concatenate([[0], np.cumsum(targets >> np.uint64(48))]).astype(np.int64)
    out = []
    for b0, b1 in ffdist.shard_bins(sizes, world):
        lo, hi = int(np.searchsorted(binidx, b0, side="left")), int(np.searchsorted(binidx, b1, side="left"))
        out.append((lo, hi, int(poff[lo]), int(poff[hi])))
    return out


def assert_reduced_equals(summ, full):
    s = full.summaries
    for f in INTS:
        assert np.array_equal(s[f], summ[f]), f
    assert np.array_equal(s["cfd_max"], summ["cfd_max"]) and np.array_equal(s["jost_max"], summ["jost_max"])
    for f in ("cfd_sum", "hsu_sum", "jost_sum"):   # added shard by shard: the same value up to the association of the additions
        assert np.abs(s[f] - summ[f]).max() <= 1e-9, f


@pytest.mark.parametrize("world,max_ot", [(2, 40), (3, 15), (5, 2000), (8, 25)])
def test_shards_on_one_device_through_the_copy_transport(capi, case, world, max_ot):
    odb, targets, positions, guides, sizes = case
    ctxs = []
    try:
        for lo, hi, plo, phi in shard_slices(targets, sizes, world):
            c = capi.Context(3)
            c.load_soa(targets[lo:hi], positions[plo:phi])
            ctxs.append(c)
        with capi.Comm.local(ctxs) as comm:
            assert comm.transport == "copy" and comm.world == world and comm.first_shard == 0
            summ = comm.discover(guides, 4, max_ot, jost=True)
            lists = [comm.shard_lists(i, jost=True) for i in range(world)]
            again = comm.discover(guides, 4, max_ot, jost=True)          # buffers reused: same answer
            assert again.tobytes() == summ.tobytes()
            tm = comm.timings()
            assert tm["scan_ms"] > 0 and tm["exchange_ms"] > 0
        with capi.Context(3) as full_ctx:
            full_ctx.load_soa(targets, positions)
            full = full_ctx.discover(guides, 4, max_ot, jost=True)
    finally:
        for c in ctxs:
            c.close()
    assert_reduced_equals(summ, full)
    G = len(guides)
    for g in range(G):   # the per-shard lists, cut off with the prior the exchange left on the device, concatenate to the unsharded list
        merged = np.concatenate([l.hits(g) for l in lists])
        assert np.array_equal(merged, full.hits(g)), g
    assert sum(int(l.n_positions) for l in lists) == int(full.n_positions)
    if max_ot < 2000:
        spans = np.array([sum(1 for l in lists if len(l.hits(g))) for g in range(G)])
        assert (spans >= 2).sum() > 0 and 0 < int(full.summaries["overflow"].sum()) < G


def test_one_rank_rccl_communicator(capi, case):
    """ncclCommInitRank with world 1: every collective of the exchange runs through librccl on the context's stream"""
    odb, targets, positions, guides, sizes = case
    uid = capi.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with capi.Context(3) as ctx:
        ctx.load_soa(targets, positions)
        with capi.Comm.rank(ctx, 0, 1, uid) as comm:
            assert comm.transport == "rccl-rank" and comm.world == 1
            summ = comm.discover(guides, 4, 37, jost=True)
            lists = comm.shard_lists(0, positions=False, hit_scores=False)
        full = ctx.discover(guides, 4, 37, jost=True)
    assert summ.tobytes() == full.summaries.tobytes()        # one shard: nothing is re-associated, bit for bit
    assert np.array_equal(lists.hit_targets, full.hit_targets)


def test_reduced_summaries_into_page_locked_and_into_pageable_memory(capi, case):
    """summaries_out of ffh_discover_sharded may be any host memory: a page-locked buffer from ffh_host_alloc (capi.host_summaries, what
    Comm.discover takes by default: copied straight into) or a pageable numpy array (staged by the runtime) -- the same bytes"""
    odb, targets, positions, guides, sizes = case
    pinned = capi.host_summaries(len(guides))
    assert pinned.dtype == capi.SUMMARY_DTYPE and len(pinned) == len(guides) and not pinned.tobytes().strip(b"\0")
    pageable = np.zeros(len(guides), dtype=capi.SUMMARY_DTYPE)
    with capi.Context(3) as ctx:
        ctx.load_soa(targets, positions)
        with capi.Comm.local([ctx]) as comm:
            g = np.ascontiguousarray(guides).view(np.uint64)
            comm._discover(g.ctypes.data, len(g), 4, 37, True, True, pinned)
            comm._discover(g.ctypes.data, len(g), 4, 37, True, True, pageable)
        full = ctx.discover(guides, 4, 37, jost=True)
    assert pinned.tobytes() == pageable.tobytes() == full.summaries.tobytes()
    view = pinned[3:5]
    del pinned                                                   # (the block lives as long as any view of it)
    assert view.tobytes() == full.summaries[3:5].tobytes()


def test_exchange_alone_after_the_callers_own_scans(capi, case):
    odb, targets, positions, guides, sizes = case
    ctxs = []
    try:
        for lo, hi, plo, phi in shard_slices(targets, sizes, 3):
            c = capi.Context(3)
            c.load_soa(targets[lo:hi], positions[plo:phi])
            ctxs.append(c)
        with capi.Comm.local(ctxs) as comm:
            with pytest.raises(capi.FlashFryHipError, match="not been scanned"):
                comm.exchange(len(guides), 30)
            with pytest.raises(capi.FlashFryHipError, match="has not run"):
                comm.shard_lists(0)
            for c in ctxs:
                c.scan(guides, 4)
            a = comm.exchange(len(guides), 30)
            b = comm.discover(guides, 4, 30)
            assert a.tobytes() == b.tobytes()
            with pytest.raises(capi.FlashFryHipError):
                comm.shard_lists(7)
    finally:
        for c in ctxs:
            c.close()


def test_cli_devices_flag_goes_through_the_library_exchange(capi, case, tmp_path):
    """flashfry-hip discover --devices 0,0,0: the CLI's multi-GPU traverser is ffh_comm_create_local + ffh_discover_sharded; its table
    must be byte-identical to the single-device run"""
    import subprocess
    from flashfry_amd import _build, synth
    odb, targets, positions, guides, sizes = case
    cli = _build.build_cli()
    db = str(tmp_path / "db")
    capi.write_database(db, 3, targets, positions, synth.CONTIGS_24)
    fa = tmp_path / "guides.fa"
    with open(fa, "w") as f:
        for i, g in enumerate(guides[:120]):
            s = "".join("ACGT"[(int(g) >> (2 * (22 - k))) & 3] for k in range(23))
            f.write(">g%d\n%s\n" % (i, s))
    outs = []
    for extra in ([], ["--devices", "0,0,0"]):
        out = str(tmp_path / ("out%d.tsv" % len(outs)))
        r = subprocess.run([cli, "discover", "--database", db, "--fasta", str(fa), "--output", out, "--maxMismatch", "4", "--maximumOffTargets", "25",
                            "--positionOutput"] + extra, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 1000


def test_a_shard_that_was_not_scanned_fails_the_exchange_on_every_shard_instead_of_hanging(capi, case):
    """a failing shard used to return before the collectives, leaving its peers inside them.  Now it takes part in the
    all-gather with a status record and every shard's caller gets the error (here: the exchange-alone entry point with one of two
    shards never scanned)."""
    odb, targets, positions, guides, sizes = case
    ctxs = []
    try:
        for lo, hi, plo, phi in shard_slices(targets, sizes, 2):
            c = capi.Context(3)
            c.load_soa(targets[lo:hi], positions[plo:phi])
            ctxs.append(c)
        ctxs[0].scan(guides, 4)
        with capi.Comm.local(ctxs) as comm:
            with pytest.raises(capi.FlashFryHipError) as e:
                comm.exchange(len(guides), 40)
            assert "shard 1" in str(e.value) and "not been scanned" in str(e.value)
            ctxs[1].scan(guides, 4)              # ... and the communicator is still usable
            summ = comm.exchange(len(guides), 40)
        with capi.Context(3) as full_ctx:
            full_ctx.load_soa(targets, positions)
            full = full_ctx.discover(guides, 4, 40)
    finally:
        for c in ctxs:
            c.close()
    assert np.array_equal(full.summaries["n_hits"], summ["n_hits"]) and np.array_equal(full.summaries["overflow"], summ["overflow"])


def test_sharded_discover_splits_a_guide_set_no_shard_can_hold(capi, case, monkeypatch):
    """VERDICT r4 next 7 on the sharded path: a shard whose scan would collect more raw hits than one scan holds (FFH_RAW_HIT_LIMIT
    puts the limit at 1024 here) reports that in its status record; every rank sees it in the same exchange, halves the guide set at
    the same place and runs the halves one after the other.  The concatenated summaries are the unsharded discover's; hit lists are
    refused after a split call and come back with a call that needs none."""
    odb, targets, positions, guides, sizes = case
    world, max_ot = 3, 25
    with capi.Context(3) as full_ctx:
        full_ctx.load_soa(targets, positions)
        full = full_ctx.discover(guides, 5, max_ot, jost=True)
        small = full_ctx.discover(guides[:8], 2, max_ot, jost=True)
    monkeypatch.setenv("FFH_RAW_HIT_LIMIT", "1024")
    ctxs = []
    try:
        for lo, hi, plo, phi in shard_slices(targets, sizes, world):
            c = capi.Context(3)
            c.load_soa(targets[lo:hi], positions[plo:phi])
            ctxs.append(c)
        with capi.Comm.local(ctxs) as comm:
            summ = comm.discover(guides, 5, max_ot, jost=True)
            with pytest.raises(capi.FlashFryHipError, match="split"):
                comm.shard_lists(0)
            few = comm.discover(guides[:8], 2, max_ot, jost=True)
            lists = [comm.shard_lists(i) for i in range(world)]
    finally:
        for c in ctxs:
            c.close()
    monkeypatch.delenv("FFH_RAW_HIT_LIMIT")
    assert_reduced_equals(summ, full)
    assert_reduced_equals(few, small)
    for g in range(8):
        assert np.array_equal(np.concatenate([l.hits(g) for l in lists]), small.hits(g))
