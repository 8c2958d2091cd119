"""Config C5 of BASELINE.json at its stated shape: Cas12a TTTV, <= 3 mismatches + one bulge, 10 000 guides
against 1.0e8 TTTN targets (hg38 holds ~1.1e8 TTTN sites), on one GPU and over eight bin shards.  The reference has no bulge search
(SURVEY.md section 8f-4): parity is unpinned by construction, the checks are (a) the seeded search == the brute-force scan of every
pair for 320 guides, (b) a second, differently structured checker (cumulated diagonals, tests/test_gpu_parity.py) on a pair sample
in both directions, (c) eight shards concatenated == the unsharded search."""
import numpy as np
import pytest

from flashfry_amd import synth

pytestmark = pytest.mark.gpu

T_C5 = int(1.0e8)
G_C5 = 10000
FIELDS = ("guide_offsets", "hit_targets", "hit_mismatches", "hit_bulge_type", "hit_bulge_position")


def _near_copies(rng, guides40, copies):
    """near-copies of the guides' protospacers: substitutions, RNA bulges (a guide base missing), DNA bulges (an extra target base)"""
    out = []
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
            out.append(v)
    return np.array(out, dtype=np.uint64)


def test_c5_cas12a_tttv_bulge_search_at_hg38_scale_and_over_eight_shards():
    import torch
    from flashfry_amd import capi, dist as ffdist
    from tests.test_gpu_parity import _bulge_by_prefix_suffix_sums
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(55)
    g40 = (synth.splitmix64(0xC7, torch.arange(G_C5, device=dev)) & ((1 << 40) - 1)).cpu().numpy().view(np.uint64)
    planted = _near_copies(rng, g40[:320], 10)
    mer = torch.unique(torch.cat([synth.splitmix64(0xC5, torch.arange(T_C5, device=dev)) & ((1 << 40) - 1),
                                  torch.from_numpy(planted.view(np.int64)).to(dev)]))
    pam_n = synth.splitmix64(0xC6, mer) & 3
    seq = (0b111111 << 42) | (pam_n << 40) | mer
    binkey = (seq >> 26) & 0x3FFF                      # the 7 bases after the 5' PAM (crispr/BinWriter.scala:58-64)
    seq = seq[torch.argsort(binkey * (1 << 48) + seq)]  # bin, then sequence: database order of a 5'-PAM enzyme
    targets = (seq | (1 << 48)).contiguous()
    T = int(targets.shape[0])
    positions = torch.arange(T, device=dev, dtype=torch.int64)
    guides = g40 | np.uint64(0b11111100 << 40) | np.uint64(1 << 48)
    with capi.Context(1) as ctx:
        torch.cuda.synchronize()
        ctx.load_soa_device(targets.data_ptr(), T, positions.data_ptr(), T)
        res = ctx.discover_bulge(guides, 3, 1, tttv=True)
        bf = ctx.discover_bulge(guides[:320], 3, 1, tttv=True, brute_force=True)
    # (a) seeded == every pair, for the guides that have planted near-copies of every alignment kind
    n = int(res.guide_offsets[320])
    assert np.array_equal(bf.guide_offsets, res.guide_offsets[:321])
    for f in FIELDS[1:]:
        assert np.array_equal(getattr(bf, f), getattr(res, f)[:n]), f
    assert {0, 1, 2} <= set(bf.hit_bulge_type.tolist()) and n > 1000 and res.n_hits > 100000
    # (b) the independent checker on a pair sample, both directions: every hit of the sampled guides must be a hit of the checker
    # with the same alignment, and no target of a 400 000-target random sample that the checker accepts may be missing
    shifts = (2 * (19 - np.arange(20))).astype(np.uint64)
    sample_idx = torch.randperm(T, device=dev, generator=torch.Generator(device=dev).manual_seed(7))[:400000].sort().values
    sample = targets[sample_idx].cpu().numpy().view(np.uint64)
    for gi in (0, 1, 7, 100, 319, 5000, 9999):
        g_bases = ((np.uint64(g40[gi]) >> shifts) & np.uint64(3)).astype(np.int8)
        a, b = int(res.guide_offsets[gi]), int(res.guide_offsets[gi + 1])
        mine = res.hit_targets[a:b]
        both = np.unique(np.concatenate([mine, sample]))
        both = both[np.argsort(((both >> np.uint64(26)) & np.uint64(0x3FFF)) * np.uint64(1 << 48) + (both & np.uint64((1 << 48) - 1)), kind="stable")]
        t_bases = ((both[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.int8)
        best, btype, bpos = _bulge_by_prefix_suffix_sums(g_bases, t_bases, 1)
        keep = (best <= 3) & (((both >> np.uint64(40)) & np.uint64(3)) != 3)
        assert np.array_equal(both[keep], mine), gi
        assert np.array_equal(best[keep].astype(np.uint8), res.hit_mismatches[a:b]) and np.array_equal(btype[keep].astype(np.uint8), res.hit_bulge_type[a:b])
        assert np.array_equal(bpos[keep].astype(np.uint8), res.hit_bulge_position[a:b])
    # (c) eight bin shards (contiguous ranges of the 16384 bins, balanced by payload), concatenated in shard order
    binidx = ((targets >> 26) & 0x3FFF)
    per_bin = torch.bincount(binidx, minlength=1 << 14).cpu().numpy()
    first = np.concatenate([[0], np.cumsum(per_bin)])
    parts = []
    for b0, b1 in ffdist.shard_bins(per_bin * 16, 8):
        lo, hi = int(first[b0]), int(first[b1])
        with capi.Context(1) as c:
            c.load_soa_device(targets[lo:hi].data_ptr(), hi - lo, positions[lo:hi].data_ptr(), hi - lo)
            r = c.discover_bulge(guides, 3, 1, tttv=True)
        parts.append({f: getattr(r, f) for f in FIELDS})
    merged = ffdist.MergedBulgeResult(parts)
    for f in FIELDS:
        assert np.array_equal(getattr(merged, f), getattr(res, f)), f
