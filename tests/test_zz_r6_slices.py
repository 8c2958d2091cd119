"""Round 6, written while the GPU pool was closed to this repository: the tests of what could not be run when it was built.  The file
sorts behind every other test file on purpose -- `pytest -x` reaches it after the suite that rounds 1-5 validated."""
import numpy as np
import pytest

from tests.test_gpu_comm import capi, case, shard_slices  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,max_ot,n_guides", [(2, 40, 300), (3, 15, 299), (5, 2000, 300), (8, 25, 300), (8, 25, 5)])
def test_exchange_by_guide_slices_equals_the_all_gather_form(capi, case, world, max_ot, n_guides):
    """round 6: ffh_comm_set_exchange(1) -- all-to-all of the records by guide slice, every rank folds its slice, the
    priors travel back, the folded slices are all-gathered -- over the copy transport: reduced aggregates byte-identical to the
    all-gather form's, and every shard's hit list (cut off with the prior the exchange left on its device) the same.  Guide counts that
    the world size does not divide, fewer guides than shards, cut-offs that cross shard boundaries (the second round)."""
    odb, targets, positions, guides, sizes = case
    guides = guides[:n_guides]
    ctxs = []
    try:
        for lo, hi, plo, phi in shard_slices(targets, sizes, world):
            c = capi.Context(3)
            c.load_soa(targets[lo:hi], positions[plo:phi])
            ctxs.append(c)
        with capi.Comm.local(ctxs) as comm:
            ref = comm.discover(guides, 4, max_ot, jost=True).copy()
            ref_lists = [comm.shard_lists(i, jost=True) for i in range(world)]
            comm.set_exchange("slice")
            got = comm.discover(guides, 4, max_ot, jost=True).copy()
            got_lists = [comm.shard_lists(i, jost=True) for i in range(world)]
            again = comm.discover(guides, 4, max_ot, jost=True)
            assert again.tobytes() == got.tobytes()
            comm.set_exchange("gather")
            back = comm.discover(guides, 4, max_ot, jost=True)
            assert back.tobytes() == ref.tobytes()
    finally:
        for c in ctxs:
            c.close()
    assert got.tobytes() == ref.tobytes()
    for a, b in zip(ref_lists, got_lists):
        assert np.array_equal(a.guide_offsets, b.guide_offsets) and np.array_equal(a.hit_targets, b.hit_targets) and a.summaries.tobytes() == b.summaries.tobytes()
    if max_ot < 2000 and n_guides > 100:
        assert 0 < int(ref["overflow"].sum()) < n_guides
