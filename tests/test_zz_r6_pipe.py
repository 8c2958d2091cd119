"""Several discover calls in flight against one resident database (round 6, VERDICT r5 missing 6 / next 6): ffh_ctx_share_db and ffh_pipe_*.
A sharing context scans the owner's database through aliases of its device memory; every scan runs the code a lone context runs, so
every result must equal, byte for byte, what sequential ffh_discover calls on the owner return."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from flashfry_amd import capi as c
    c.load_library()
    return c


@pytest.fixture(scope="module")
def case(oracle):
    from tests.test_gpu_parity import dense_case
    odb, targets, positions, guides = dense_case(oracle, n_random=300000, n_guides=480, n_dense=60, variants=150, seed=77)
    return odb, targets, positions, guides


def same(a, b, lists=True):
    assert a.summaries.tobytes() == b.summaries.tobytes()
    if lists:
        assert np.array_equal(a.guide_offsets, b.guide_offsets) and np.array_equal(a.hit_targets, b.hit_targets)
        assert np.array_equal(a.hit_mismatches, b.hit_mismatches) and np.array_equal(a.positions, b.positions)


def test_pipe_results_equal_sequential_calls(capi, case, oracle):
    """eight distinct guide batches (sizes differ, cut-offs bite, <= 3 .. 5 mismatches mixed) through a two-lane pipe, collected out of
    order: each result equals the sequential call's; the first batch also equals the oracle"""
    from tests.helpers import assert_same_hits
    odb, targets, positions, guides = case
    batches = [(guides[i * 60:(i + 1) * 60 - (i % 3)], 3 + i % 3, [40, 2000, 15][i % 3]) for i in range(8)]
    with capi.Context(3) as ctx:
        ctx.load_soa(targets, positions)
        seq = [ctx.discover(g, mm, ot, jost=True) for g, mm, ot in batches]
        seq_sum = [ctx.discover(g, mm, ot, summaries_only=True) for g, mm, ot in batches]
        with ctx.pipe(2) as pipe:
            assert pipe.lanes == 2
            tickets = [pipe.submit(g, mm, ot, jost=True) for g, mm, ot in batches]
            tickets_sum = [pipe.submit(g, mm, ot, summaries_only=True) for g, mm, ot in batches]
            got_sum = {t: pipe.wait(t) for t in reversed(tickets_sum)}
            got = {t: pipe.wait(t) for t in reversed(tickets)}
            with pytest.raises(capi.FlashFryHipError):
                pipe.wait(tickets[0])            # collected already
        for t, r in zip(tickets, seq):
            same(got[t], r)
        for t, r in zip(tickets_sum, seq_sum):
            same(got_sum[t], r, lists=False)
        again = ctx.discover(*batches[0][:1], batches[0][1], batches[0][2], jost=True)   # the owner is its own again after the pipe
        same(again, seq[0])
    g, mm, ot = batches[0]
    assert_same_hits(seq[0], odb.discover(g, mm, ot))


def test_two_host_threads_on_a_shared_database(capi, case):
    """ffh_ctx_share_db alone: the owner and a sharing context driven from two host threads at the same time, 20 calls each (bounded and
    unbounded, lists and aggregates) -- every answer the sequential one; the owner refuses to load or rebuild while it is shared"""
    odb, targets, positions, guides = case
    ga, gb = guides[:240], guides[240:]
    with capi.Context(3) as ctx:
        ctx.load_soa(targets, positions)
        ref_a, ref_b = ctx.discover(ga, 4, 60, jost=True), ctx.discover(gb, 4, 60, jost=True)
        ref_a5 = ctx.discover(ga, 5, 2000, summaries_only=True)
        other = ctx.share()
        try:
            with pytest.raises(capi.FlashFryHipError) as e:
                ctx.load_soa(targets, positions)
            assert "shared" in str(e.value)
            with pytest.raises(capi.FlashFryHipError):
                other.load_soa(targets, positions)
            with pytest.raises(capi.FlashFryHipError):
                other.share()
            errors = []

            def run(c, g, ref, bounding):
                try:
                    c.set_bounding(bounding)
                    for k in range(20):
                        r = c.discover(g, 4, 60, jost=True) if k % 2 == 0 else c.discover(g, 4, 60, summaries_only=True, jost=True)
                        same(r, ref, lists=k % 2 == 0)
                except Exception as ex:   # noqa: BLE001
                    errors.append(repr(ex))
            ta = threading.Thread(target=run, args=(ctx, ga, ref_a, 1))
            tb = threading.Thread(target=run, args=(other, gb, ref_b, 0))
            ta.start(); tb.start(); ta.join(); tb.join()
            assert not errors, errors
            same(other.discover(ga, 5, 2000, summaries_only=True), ref_a5, lists=False)   # another maxMismatch: the sharing context picks (and builds) its own images
        finally:
            other.close()
        ctx.load_soa(targets[:1000], positions[:int((targets[:1000] >> np.uint64(48)).sum())])   # not shared any more: loads again
