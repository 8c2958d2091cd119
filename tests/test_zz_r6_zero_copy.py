"""Round 6, written while the GPU pool was closed to this repository: FFH_LIST_ZERO_COPY (see tests/test_zz_r6_slices.py for why the
file sorts last)."""
import numpy as np
import pytest

from tests.helpers import assert_same_hits
from tests.test_gpu_configs import capi  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_list_arrays_stored_by_the_kernels_themselves_deliver_the_same_result(capi, oracle, monkeypatch):
    """FFH_LIST_ZERO_COPY=1 (round 6; off until measured): a list-delivering finalize lets k_score_hits store the hit targets, mismatches
    and per-hit scores, and k_gather_positions the positions, straight into the result's page-locked block instead of copying device
    arrays afterwards.  Every array must be what the copying form delivers: with and without positions / per-hit scores, cut-off far and
    biting, also through ffh_finalize after a scan of its own and on a repeat-structured genome under the bounded scan."""
    from flashfry_amd import synth
    from tests.test_gpu_parity import dense_case
    odb, t, p, g = dense_case(oracle, n_random=150_000, n_guides=600, n_dense=50, variants=150, seed=67)
    db = synth.make_repeat_database(300_000, seed=5, repeat_fraction=0.4)
    rg = synth.as_u64(synth.make_guides_from_database(db, 200, seed=6))
    rt, rp = synth.as_u64(db["targets"]), synth.as_u64(db["positions"])
    kws = ((), (("positions", False), ("hit_scores", False)), (("hit_scores", False),))

    def run():
        out = {}
        with capi.Context(3) as ctx:
            ctx.load_soa(t, p)
            for mo in (2000, 30):
                for kw in kws:
                    out[("dense", mo, kw)] = ctx.discover(g, 4, mo, jost=True, **dict(kw))
            ctx.scan(g, 3)
            out[("two-step", 0, ())] = ctx.finalize(2000, jost=True)
        with capi.Context(3) as ctx:
            ctx.load_soa(rt, rp)
            ctx.set_bounding(1)
            out[("repeats", 60, ())] = ctx.discover(rg, 4, 60)
        return out
    plain = run()
    monkeypatch.setenv("FFH_LIST_ZERO_COPY", "1")
    direct = run()
    monkeypatch.delenv("FFH_LIST_ZERO_COPY")
    for key, want in plain.items():
        got, kw = direct[key], dict(key[2])
        assert got.n_hits == want.n_hits and got.n_positions == want.n_positions, key
        assert got.summaries.tobytes() == want.summaries.tobytes(), key
        for name in ("guide_offsets", "hit_targets", "hit_mismatches"):
            assert np.array_equal(getattr(got, name), getattr(want, name)), (key, name)
        if kw.get("positions", True):
            assert np.array_equal(got.positions, want.positions) and np.array_equal(got.pos_offsets, want.pos_offsets), key
        if kw.get("hit_scores", True):
            assert np.array_equal(np.isnan(got.hit_cfd), np.isnan(want.hit_cfd)) and np.array_equal(np.nan_to_num(got.hit_cfd), np.nan_to_num(want.hit_cfd)), key
    assert_same_hits(plain[("dense", 30, ())], odb.discover(g, 4, 30))
