#!/usr/bin/env python3
"""tools/slab_sim.py -- how many raw hits a bounded scan of the repeat-structured workload would record if the hits of a guide that reaches
maximumOffTargets INSIDE a slab were dropped beyond the sub-range of the slab's index span in which it reaches it (S equal sub-ranges
per slab), against the slab-granular rule (S = 1: what the scan does).  A sample of the guides is scanned unbounded with every hit
delivered; the rest is arithmetic on the hits' database indices and position counts."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=float, default=3.0e8)
    ap.add_argument("--guides", type=int, default=100000)
    ap.add_argument("--sample", type=int, default=3000)
    ap.add_argument("--max-offtargets", type=int, default=2000)
    args = ap.parse_args()
    import torch
    from flashfry_amd import capi, synth
    dev = torch.device("cuda:0")
    db = synth.make_repeat_database(int(args.targets), seed=synth.DB_SEED + 99, device=dev)
    guides = synth.make_guides_from_database(db, args.guides, device=dev).cpu().numpy().view(np.uint64)
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(len(guides), args.sample, replace=False))
    T = db["T"]
    seq_mask = (1 << 48) - 1
    with capi.Context(3, device=0) as ctx:
        torch.cuda.synchronize()
        ctx.load_soa_device(db["targets"].data_ptr(), T, db["positions"].data_ptr(), db["P"])
        ctx.set_bounding(0)
        res = ctx.discover(guides[pick], 4, 2 ** 31 - 1, positions=False, hit_scores=False)
        off = res.guide_offsets.astype(np.int64)
        ht = torch.from_numpy(res.hit_targets.view(np.int64).copy()).to(dev)
    seqs = (db["targets"] & seq_mask)
    idx = torch.searchsorted(seqs, ht & seq_mask).cpu().numpy()
    cnt = ((ht >> 48) & 0xFFFF).cpu().numpy().astype(np.int64)
    # the slabs' index bounds: first-three-bases ranks 0, 1, 4, 12, 24, 40, 64 of 64 (ffh_scan.inc kSlabRank); bases lead the 46-bit sequence
    top = (seqs >> 40).cpu().numpy() if False else None
    rank_cut = [0, 1, 4, 12, 24, 40, 64]
    # first three bases = the top 6 bits of the 46-bit planar?  the database is in sequence order: take the cuts from the rank of the leading bases
    lead = ((db["targets"] & seq_mask) >> 40)
    nlead = int(lead.max().item()) + 1
    # generic: cut the index space where the leading 6 bits change (64 equal classes of the leading field)
    keyspace = 1 << 46   # 23 bases (k_slab_cuts: rank = (t >> 40) & 63)
    cuts = [int(torch.searchsorted(seqs, torch.tensor([min(keyspace - 1, r * keyspace // 64)], device=dev, dtype=seqs.dtype)).item()) for r in rank_cut[:-1]] + [T]
    out = {"sample": int(args.sample), "raw_unbounded": int(len(idx)), "slab_cuts": cuts}
    limit = args.max_offtargets
    for S in (1, 4, 16, 64, 256):
        raw = kept = 0
        for g in range(len(pick)):
            a, b = off[g], off[g + 1]
            ii, cc = idx[a:b], cnt[a:b]
            cum = np.cumsum(cc)
            nk = int(np.searchsorted(cum - cc, limit, side="left"))   # kept: running total BEFORE the hit < limit
            nk = min(nk, len(ii))
            kept += nk
            if nk == len(ii):
                raw += nk
                continue
            cross = ii[nk - 1] if nk else ii[0]            # index of the hit that reaches the limit
            k = int(np.searchsorted(cuts, cross, side="right")) - 1
            lo, hi = cuts[k], cuts[k + 1]
            sub = min(S - 1, int((cross - lo) * S // max(hi - lo, 1)))
            end = lo + (hi - lo) * (sub + 1) // S          # records with index < end are recorded
            raw += int(np.searchsorted(ii, end, side="left"))
        out["S=%d" % S] = {"raw": raw, "kept": kept, "raw_over_kept": round(raw / max(kept, 1), 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
