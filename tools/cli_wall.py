#!/usr/bin/env python3
"""Process wall time of the drop-in CLI (argv to closed output file) at chr22 scale -- SURVEY.md section 8d "t_discover".

Builds a seeded random genome (default 50 Mb in 4 contigs, ~6e6 NGG sites: chr22 has 51 Mb), indexes it with
`flashfry-hip index`, then times `discover` for 1 guide (the README's EMX1 guide, config C1) and 1 000 random guides
(config C2), each (a) right after the page cache was dropped (when /proc/sys/vm/drop_caches is writable) and (b) warm,
then `score`. Everything is the product path; nothing here reads the oracle or the reference.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "flashfry_amd", "bin", "flashfry-hip")


def write_genome(path, mbases, contigs, seed):
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "wb") as f:
        for c in range(contigs):
            n = int(mbases * 1e6 / contigs)
            f.write(b">chrS%d synthetic\n" % (c + 1))
            for a in range(0, n - n % 60, 60 << 20):  # 60-column lines, written in slabs
                m = min(60 << 20, n - n % 60 - a)
                rows = np.empty((m // 60, 61), dtype=np.uint8)
                rows[:, :60] = lut[rng.integers(0, 4, m, dtype=np.uint8)].reshape(-1, 60)
                rows[:, 60] = 10
                f.write(rows.tobytes())


def write_guides(path, n, seed):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for _ in range(n):
            s = "".join("ACGT"[i] for i in rng.integers(0, 4, 21)) + "GG"
            f.write(">random%s\n%s\n" % (s, s))


def drop_caches():
    try:
        subprocess.run(["sync"], check=False)
        with open("/proc/sys/vm/drop_caches", "w") as f:
            f.write("3\n")
        return True
    except OSError:
        return False


def timed(cmd, key="comparisons"):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    dt = time.perf_counter() - t0
    if p.returncode:
        raise SystemExit("FAILED %s\n%s" % (" ".join(cmd), p.stderr[-2000:]))
    detail = [l for l in p.stderr.splitlines() if key in l or "Database load" in l or "Host stages" in l or "[ffh ingest]" in l]
    return dt, " | ".join(detail)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbases", type=float, default=50.0)
    ap.add_argument("--contigs", type=int, default=4)
    ap.add_argument("--workdir", default="/tmp/ff_cli_wall")
    ap.add_argument("--big-guides", type=int, default=0, help="also time discover with this many random guides (config C3: 100000), without --positionOutput like the paper's runs")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cli_wall.json"))
    args = ap.parse_args()
    os.makedirs(args.workdir, exist_ok=True)
    w = args.workdir
    fa, db = os.path.join(w, "genome.fa"), os.path.join(w, "genome_cas9ngg")
    write_genome(fa, args.mbases, args.contigs, 20240922)
    with open(os.path.join(w, "emx1.fa"), "w") as f:
        f.write(">EMX1\nGAGTCCGAGCAGAAGAAGAAGGG\n")
    write_guides(os.path.join(w, "g1000.fa"), 1000, 0x6D1DE5)
    rows = {}
    rows["genome_mbases"] = args.mbases
    rows["index_s"], rows["index_detail"] = timed([CLI, "index", "--reference", fa, "--database", db, "--enzyme", "spcas9ngg", "--tmpLocation", w], key="Wrote")
    rows["database_bytes"] = os.path.getsize(db)
    for name, guides in (("C1_1_guide", "emx1.fa"), ("C2_1000_guides", "g1000.fa")):
        out = os.path.join(w, name + ".output")
        cmd = [CLI, "discover", "--database", db, "--fasta", os.path.join(w, guides), "--output", out, "--positionOutput", "--maxMismatch", "4"]
        cold = drop_caches()
        t_cold, _ = timed(cmd)
        warm = [timed(cmd) for _ in range(3)]
        t_warm, detail = min(warm)
        t_score, _ = timed([CLI, "score", "--input", out, "--output", os.path.join(w, name + ".scored"), "--database", db, "--scoringMetrics",
                            "doench2016cfd,hsu2013,minot,dangerous"])
        rows[name] = {"discover_first_s": t_cold, "page_cache_dropped": cold, "discover_warm_s": t_warm, "score_s": t_score, "detail": detail,
                      "output_bytes": os.path.getsize(out)}
    if args.big_guides:
        gpath, out = os.path.join(w, "gbig.fa"), os.path.join(w, "C3.output")
        write_guides(gpath, args.big_guides, 0xC3)
        cmd = [CLI, "discover", "--database", db, "--fasta", gpath, "--output", out, "--maxMismatch", "4"]
        runs = [timed(cmd) for _ in range(2)]
        t, detail = min(runs)
        t_score, _ = timed([CLI, "score", "--input", out, "--output", os.path.join(w, "C3.scored"), "--database", db, "--scoringMetrics", "doench2016cfd,hsu2013"])
        rows["C3_%d_guides" % args.big_guides] = {"discover_warm_s": t, "score_s": t_score, "detail": detail, "output_bytes": os.path.getsize(out)}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rows, f, indent=1)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
