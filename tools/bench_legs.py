"""Optional legs of bench.py: they run only when the box offers what they need and skip cleanly otherwise.

  jvm_leg      `java` on PATH and $FLASHFRY_JAR set: the REAL reference (`java -Xmx15g -jar flashfry.jar discover ...`, pinned to one core
               with taskset and timed with its max RSS exactly as paper/tools/flashfry_off_target.cwl:16-40 and
               paper/run_timing_collection.py:9-24 do) on a database file written in the reference's own format from a bounded sample
               of the bench's synthetic database -> cpu_baseline.kind "reference".
  fasta_leg    $FF_GENOME_FASTA set: `index` + `discover` on a real genome (plain or .gz FASTA) through the product path, guides sampled
               from the genome itself as real libraries are.
  pin_to_core / max_rss_kib: the single-thread CPU leg runs pinned (os.sched_setaffinity) and reports ru_maxrss.
"""
import gzip
import os
import resource
import shutil
import subprocess
import tempfile
import time

import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def pin_to_core():
    """pins the calling thread to the first core of its affinity mask; returns (core, restore())"""
    try:
        old = os.sched_getaffinity(0)
        core = min(old)
        os.sched_setaffinity(0, {core})
        return core, (lambda: os.sched_setaffinity(0, old))
    except (AttributeError, OSError):
        return None, (lambda: None)


def max_rss_kib():
    return int(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)  # KiB on Linux, the unit /usr/bin/time -v prints


def longs_to_strings(longs, length):
    """2-bit longs (first base most significant) -> list of ASCII sequences of `length` bases"""
    v = np.asarray(longs, dtype=np.uint64)
    sh = (2 * (length - 1 - np.arange(length))).astype(np.uint64)
    codes = ((v[:, None] >> sh[None, :]) & np.uint64(3)).astype(np.uint8)
    return [row.tobytes().decode() for row in BASES[codes]]


def write_guides_fasta(path, guides, length=23):
    with open(path, "w") as f:
        for i, s in enumerate(longs_to_strings(guides, length)):
            f.write(">guide%d\n%s\n" % (i, s))


def jvm_leg(capi, targets, positions, contigs, guides, max_mm, max_ot, budget_note):
    """times the reference jar on (targets, positions) written as a reference-format database; None when java or the jar is absent"""
    jar = os.environ.get("FLASHFRY_JAR")
    java = shutil.which("java")
    if not jar or not java or not os.path.exists(jar):
        return None
    work = tempfile.mkdtemp(prefix="ffh_jvm_")
    try:
        db = os.path.join(work, "db")
        capi.write_database(db, 3, targets, positions, contigs)
        fa = os.path.join(work, "guides.fasta")
        write_guides_fasta(fa, guides)
        out = os.path.join(work, "out.txt")
        cmd = [java, "-Xmx15g", "-jar", jar, "discover", "--database", db, "--fasta", fa, "--output", out,
               "--maxMismatch", str(max_mm), "--maximumOffTargets", str(max_ot)]
        core = None
        if shutil.which("taskset"):
            core = min(os.sched_getaffinity(0))
            cmd = ["taskset", "-c", str(core)] + cmd
        timer = "/usr/bin/time"
        if os.path.exists(timer):
            cmd = [timer, "-v"] + cmd
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=3600)
        dt = time.perf_counter() - t0
        rss = None
        for line in r.stderr.splitlines():
            if "Maximum resident set size (kbytes)" in line:
                rss = int(line.split("):")[1])
        if r.returncode != 0 or not os.path.exists(out):
            return {"value": None, "unit": "guide*target comparisons/s", "cores": 1, "kind": "reference", "sample": "java run failed: " + r.stderr[-400:]}
        return {"value": len(guides) * len(targets) / dt, "unit": "guide*target comparisons/s", "cores": 1, "kind": "reference",
                "sample": "java -Xmx15g -jar $FLASHFRY_JAR discover, pinned to core %s, %d guides vs %d targets (%s), whole process incl. JVM start and "
                          "database read: %.1f s" % (core, len(guides), len(targets), budget_note, dt),
                "seconds": dt, "max_rss_kib": rss}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def read_fasta(path):
    """[(name, bytes)] of a plain or gzip-compressed FASTA"""
    op = gzip.open if path.endswith(".gz") else open
    name, chunks, out = None, [], []
    with op(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if name is not None:
                    out.append((name, b"".join(chunks)))
                name, chunks = line[1:].split()[0].decode(), []
            else:
                chunks.append(line.strip())
    if name is not None:
        out.append((name, b"".join(chunks)))
    return out


def sample_ngg_sites(contigs, n, seed):
    """n forward-strand 20-mer + NGG sites (ACGT only) drawn at random offsets of random contigs (length-weighted), as guide longs"""
    rng = np.random.default_rng(seed)
    lens = np.array([len(s) for _, s in contigs], dtype=np.float64)
    if lens.sum() < 23:
        return np.zeros(0, dtype=np.uint64)
    code = np.full(256, 255, dtype=np.uint8)
    for k, ch in enumerate(b"ACGT"):
        code[ch] = k
        code[ch + 32] = k  # soft-masked (lower-case) bases count as bases (ReferenceEncoder upper-cases the contig)
    out = []
    tries = 0
    while len(out) < n and tries < 50 * n + 1000:
        tries += 1
        c = int(rng.choice(len(contigs), p=lens / lens.sum()))
        seq = contigs[c][1]
        if len(seq) < 23 + 200:
            continue
        o = int(rng.integers(0, len(seq) - 223))
        win = code[np.frombuffer(seq[o:o + 223], dtype=np.uint8)]
        for k in range(200):
            w = win[k:k + 23]
            if w[21] == 2 and w[22] == 2 and w.max() < 4:
                v = 0
                for b in w:
                    v = (v << 2) | int(b)
                out.append(v | (1 << 48))
                break
    return np.array(out, dtype=np.uint64)


def fasta_leg(capi, n_guides, max_mm, max_ot, device=0, seed=12345):
    """index + discover on $FF_GENOME_FASTA; None when the variable is not set"""
    path = os.environ.get("FF_GENOME_FASTA")
    if not path or not os.path.exists(path):
        return None
    work = tempfile.mkdtemp(prefix="ffh_fasta_")
    try:
        t0 = time.perf_counter()
        contigs = read_fasta(path)
        t_read = time.perf_counter() - t0
        db = os.path.join(work, "db")
        t0 = time.perf_counter()
        st = capi.index_contigs(db, 3, contigs, device=device)
        t_index = time.perf_counter() - t0
        with capi.Context(0, device=device) as ctx:
            t0 = time.perf_counter()
            ctx.open(db)
            t_load = time.perf_counter() - t0
            info = ctx.info()
            # guides sampled from the genome itself, as real libraries are: forward-strand N20-NGG sites at random places
            guides = sample_ngg_sites(contigs, n_guides, seed)
            if len(guides) == 0:
                return {"fasta": os.path.basename(path), "index_seconds": t_index, "note": "no NGG site found for the guides"}
            n_guides = len(guides)
            ctx.discover(guides, max_mm, max_ot, summaries_only=True)  # warm-up
            t0 = time.perf_counter()
            res = ctx.discover(guides, max_mm, max_ot, summaries_only=True)
            t_disc = time.perf_counter() - t0
            tm = ctx.timings()
            return {"fasta": os.path.basename(path), "bases": int(st.n_bases), "targets": int(info.n_targets), "positions": int(info.n_positions),
                    "read_seconds": t_read, "index_seconds": t_index, "load_seconds": t_load, "guides": int(n_guides),
                    "discover_ms": t_disc * 1e3, "compare_ms": tm.compare_ms, "raw_hits": int(tm.n_raw_hits),
                    "overflowed_guides": int(res.summaries["overflow"].sum()),
                    "comparisons_per_s": n_guides * int(info.n_targets) / t_disc}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def c2_leg(torch, capi, synth, device, max_mm, max_ot, cli=None, steps=30):
    """BASELINE.json configs[1] and SURVEY.md section 8d's chr22-scale line: 1 000 random NGG guides against a chr22-scale synthetic
    database (4.5e6 unique targets), <= max_mm mismatches, one GPU.  (a) the resident step: aggregates only, and with the hit lists and
    positions delivered; (b) `t_discover`: the drop-in CLI from argv to the closed output file on the same database WRITTEN TO DISK in
    the reference's format (BGZF body + text header), first run (the file was just written: the kernel's page cache holds it, the
    process pays header parse, BGZF member directory, device inflate, block decode, scan images) and median of three more."""
    import subprocess
    dev = torch.device("cuda", device)
    G, T_req = 1000, int(4.5e6)
    guides_dev = synth.make_guides(G, seed=synth.GUIDE_SEED + 2, device=dev)
    db = synth.make_database(T_req, seed=synth.DB_SEED + 2, plant_guides=guides_dev, device=dev)
    guides = guides_dev.cpu().numpy().view(np.uint64)
    out = {"workload": "chr22-scale (BASELINE.json configs[1]): %d random NGG guides vs %d unique targets (%d positions), <=%d mismatches, maximumOffTargets %d, 1 GPU"
                       % (G, db["T"], db["P"], max_mm, max_ot)}
    with capi.Context(3, device=device) as ctx:
        torch.cuda.synchronize()
        ctx.load_soa_device(db["targets"].data_ptr(), db["T"], db["positions"].data_ptr(), db["P"])
        out["db_prepare_ms"] = ctx.info().prepare_ms
        for name, kw in (("ms_per_step", dict(summaries_only=True)), ("discover_with_lists_ms", dict(hit_scores=False))):
            for _ in range(3):
                ctx.discover(guides, max_mm, max_ot, **kw)
            ts = []
            for _ in range(steps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = ctx.discover(guides, max_mm, max_ot, **kw)
                ts.append((time.perf_counter() - t0) * 1e3)
            out[name] = float(np.median(ts))
        tm = ctx.timings().as_dict()
        out.update(value=G * db["T"] / (out["ms_per_step"] * 1e-3), unit="comparisons/s", raw_hits=int(tm["n_raw_hits"]), compare_ms=tm["compare_ms"],
                   plan=[tm["prefix_bases"], tm["prefix_radius"], tm["suffix_radius"]], kept_positions=int(res.summaries["ot_count"].sum()))
    if cli and os.path.exists(cli):
        work = tempfile.mkdtemp(prefix="ffh_c2_")
        try:
            dbp = os.path.join(work, "db")
            t0 = time.perf_counter()
            capi.write_database(dbp, 3, db["targets"].cpu().numpy().view(np.uint64), db["positions"].cpu().numpy().view(np.uint64), synth.CONTIGS_24)
            out["db_write_s"] = time.perf_counter() - t0
            out["db_file_bytes"] = os.path.getsize(dbp)
            fa = os.path.join(work, "guides.fasta")
            write_guides_fasta(fa, guides)
            walls = []
            for k in range(4):
                o = os.path.join(work, "out%d.tsv" % k)
                t0 = time.perf_counter()
                r = subprocess.run([cli, "discover", "--database", dbp, "--fasta", fa, "--output", o, "--maxMismatch", str(max_mm), "--maximumOffTargets", str(max_ot),
                                    "--positionOutput"], capture_output=True, timeout=600)
                walls.append(time.perf_counter() - t0)
                if r.returncode != 0:
                    out["cli_error"] = r.stderr.decode()[-300:]
                    break
            if "cli_error" not in out:
                out["discover_wall_first_s"] = walls[0]
                out["discover_wall_warm_s"] = float(np.median(walls[1:]))
                out["table_bytes"] = os.path.getsize(os.path.join(work, "out0.tsv"))
        finally:
            shutil.rmtree(work, ignore_errors=True)
    return out
