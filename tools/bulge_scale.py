#!/usr/bin/env python3
"""Config C5 at scale: Cas12a TTTV, <= 3 mismatches + one bulge over a synthetic TTTN database (default 1e8 targets, hg38 has
~1.1e8 TTTN sites per strand pair) for --guides random guides on one GPU: the seeded candidate search (the default) and, with
--brute-guides N, the brute-force scan of every pair for the first N guides (its checker; the hit arrays must agree)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=float, default=1.0e8)
    ap.add_argument("--guides", type=int, default=1000)
    ap.add_argument("--brute-guides", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bulge_scale.json"))
    args = ap.parse_args()
    import torch
    from flashfry_amd import capi, synth
    dev = torch.device("cuda", 0)
    n = int(args.targets)
    mer = torch.unique(synth.splitmix64(0xC5, torch.arange(n, device=dev)) & ((1 << 40) - 1))
    pam_n = synth.splitmix64(0xC6, mer) & 3
    seq = (0b111111 << 42) | (pam_n << 40) | mer
    binkey = (seq >> 26) & 0x3FFF                      # the 7 bases after the 5' PAM (crispr/BinWriter.scala:58-64)
    order = torch.argsort(binkey * (1 << 48) + seq)    # bin, then sequence: database order of a 5'-PAM enzyme
    targets = (seq[order] | (1 << 48)).contiguous()
    T = int(targets.shape[0])
    positions = torch.arange(T, device=dev, dtype=torch.int64)
    g = (synth.splitmix64(0xC7, torch.arange(args.guides, device=dev)) & ((1 << 40) - 1)) | (0b11111100 << 40) | (1 << 48)
    guides = g.cpu().numpy().view(np.uint64)
    out = {"targets": T, "guides": args.guides}
    with capi.Context(1) as ctx:
        torch.cuda.synchronize()
        ctx.load_soa_device(targets.data_ptr(), T, positions.data_ptr(), T)
        for label, mm, bulge, tttv in (("mismatch_only_3mm", 3, 0, True), ("3mm_plus_1_bulge_TTTV", 3, 1, True), ("3mm_plus_1_bulge_TTTN", 3, 1, False)):
            ctx.discover_bulge(guides[:8], mm, bulge, tttv=tttv)
            t0 = time.perf_counter()
            res = ctx.discover_bulge(guides, mm, bulge, tttv=tttv)
            dt = time.perf_counter() - t0
            out[label] = {"seconds": dt, "hits": int(res.n_hits), "pairs_per_s": args.guides * T / dt,
                          "by_type": [int((res.hit_bulge_type == k).sum()) for k in range(3)]}
            if args.brute_guides:
                nb = min(args.brute_guides, args.guides)
                t0 = time.perf_counter()
                bf = ctx.discover_bulge(guides[:nb], mm, bulge, tttv=tttv, brute_force=True)
                dtb = time.perf_counter() - t0
                n = int(res.guide_offsets[nb])
                same = (np.array_equal(bf.guide_offsets, res.guide_offsets[:nb + 1]) and np.array_equal(bf.hit_targets, res.hit_targets[:n]) and
                        np.array_equal(bf.hit_mismatches, res.hit_mismatches[:n]) and np.array_equal(bf.hit_bulge_type, res.hit_bulge_type[:n]) and
                        np.array_equal(bf.hit_bulge_position, res.hit_bulge_position[:n]))
                out[label]["brute_force"] = {"guides": nb, "seconds": dtb, "pairs_per_s": nb * T / dtb, "equal_to_seeded": bool(same),
                                             "seeded_speedup_per_guide": (dtb / nb) / (dt / args.guides)}
                if not same:
                    raise SystemExit("seeded bulge search differs from brute force (%s)" % label)
            print(label, json.dumps(out[label]), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
