#!/usr/bin/env python3
"""Database ingest at scale: write a synthetic database of --targets unique targets with
ffh_db_write, then time ffh_db_open -- BGZF inflate on the host cores into page-locked buffers, overlapped copies, block
decode and scan-image build on the device -- right after dropping the page cache and warm, and check that the database
loaded from the file gives the same discover result as the same arrays handed over directly."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def drop_caches():
    try:
        os.sync()
        with open("/proc/sys/vm/drop_caches", "w") as f:
            f.write("3\n")
        return True
    except OSError:
        return False


def digest(res):
    h = hashlib.sha256()
    for a in (res.guide_offsets, res.hit_targets, res.pos_offsets, res.positions, res.summaries):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=float, default=3.0e8)
    ap.add_argument("--guides", type=int, default=1000)
    ap.add_argument("--path", default="/tmp/ff_ingest/db")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ingest_scale.json"))
    args = ap.parse_args()
    import torch
    from flashfry_amd import capi, synth

    os.makedirs(os.path.dirname(args.path), exist_ok=True)
    dev = torch.device("cuda", 0)
    guides_dev = synth.make_guides(args.guides, device=dev)
    db = synth.make_database(int(args.targets), seed=synth.DB_SEED, plant_guides=guides_dev, device=dev)
    guides = guides_dev.cpu().numpy().view(np.uint64)
    out = {"targets": db["T"], "positions": db["P"], "host_cores": os.cpu_count()}
    with capi.Context(3) as ctx:
        ctx.load_soa_device(db["targets"].data_ptr(), db["T"], db["positions"].data_ptr(), db["P"])
        want = digest(ctx.discover(guides, 4, 2000))
    t_host, p_host = db["targets"].cpu().numpy(), db["positions"].cpu().numpy()
    del db
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    capi.write_database(args.path, 3, t_host, p_host, synth.CONTIGS_24)
    out["write_s"] = time.perf_counter() - t0
    out["file_bytes"] = os.path.getsize(args.path)
    del t_host, p_host
    runs = []
    for label in ("cold", "warm", "warm"):
        dropped = drop_caches() if label == "cold" else False
        with capi.Context(0) as ctx:
            t0 = time.perf_counter()
            ctx.open(args.path)
            dt = time.perf_counter() - t0
            st = ctx.load_stats().as_dict()
            info = ctx.info()
            assert (info.n_targets, info.n_positions) == (out["targets"], out["positions"])
            got = digest(ctx.discover(guides, 4, 2000))
        assert got == want, "the database loaded from the file gives a different discover result"
        runs.append(dict(st, label=label, page_cache_dropped=dropped, open_wall_s=dt))
        print(json.dumps(runs[-1]), flush=True)
    out["runs"] = runs
    out["same_discover_result_as_direct_load"] = True
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
