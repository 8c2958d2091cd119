"""The C++ host CLI (flashfry-hip index | discover | score) against the oracle's restatement of the reference CLI.
Option handling is checked without a GPU; index, discover and score compute on the device and need the GPU box."""
import os
import subprocess

import numpy as np
import pytest

from flashfry_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cli():
    return _build.build_cli()


def random_genome(path, n_contigs=3, length=60000, seed=1, width=70):
    """random contigs with some soft-masked (lower-case) stretches, N runs and a planted repeat"""
    rng = np.random.default_rng(seed)
    repeat = "".join(rng.choice(list("ACGT"), 300))
    with open(path, "w") as f:
        for c in range(n_contigs):
            s = "".join(rng.choice(list("ACGT"), length))
            s = s[:1000] + "N" * 50 + s[1050:5000] + repeat + s[5300:20000] + repeat + s[20300:30000].lower() + s[30000:]
            f.write(">chr%d test contig\t%d\n" % (c + 1, c))
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


def guide_fasta(path, genome_path, n=40, seed=2, mutate=2):
    """guides cut out of the genome (so they have on- and off-targets), lightly mutated, in the `random` module's format"""
    rng = np.random.default_rng(seed)
    seq = "".join(l.strip() for l in open(genome_path) if not l.startswith(">")).upper()
    out = []
    while len(out) < n:
        # every other guide comes out of the planted repeat (6 copies genome-wide -> it overflows a small maximumOffTargets)
        i = int(rng.integers(5000, 5277)) if len(out) % 2 else int(rng.integers(0, len(seq) - 23))
        w = seq[i:i + 23]
        if w.endswith("GG") and set(w) <= set("ACGT") and not w.startswith("CC"):
            w = list(w)
            for p in rng.choice(20, size=int(rng.integers(0, mutate + 1)), replace=False):
                w[p] = rng.choice(list("ACGT"))
            out.append("".join(w))
            if len(out) % 5 == 1 and len(out) < n:  # a sibling one base away: reciprocal off-targets of each other
                k = int(rng.integers(2, 20))
                w[k] = "ACGT"[("ACGT".index(w[k]) + 1) % 4]
                out.append("".join(w))
    with open(path, "w") as f:
        for w in out:
            f.write(">random%s\n%s\n" % (w, w))


@pytest.mark.gpu
@pytest.mark.parametrize("enzyme", ["spcas9ngg", "spcas9", "cpf1", "spcas9ngg19", "spcas9nag", "spcas919"])
def test_index_matches_oracle(cli, oracle, tmp_path, enzyme):
    fa = str(tmp_path / "genome.fa")
    random_genome(fa, seed=len(enzyme))
    a, b = str(tmp_path / "db_cli"), str(tmp_path / "db_oracle")
    subprocess.check_call([cli, "index", "--reference", fa, "--database", a, "--enzyme", enzyme, "--tmpLocation", str(tmp_path)], stderr=subprocess.DEVNULL)
    assert oracle.lib.ffo_index_fasta(fa.encode(), b.encode(), enzyme.encode(), 7) == 0, oracle.error()
    ha, hb = open(a + ".header").read().split("\n"), open(b + ".header").read().split("\n")
    assert ha[:4] == hb[:4]
    # bin sizes, target counts and contig lines are identical; the BGZF virtual offsets only depend on the block size used
    assert [l.split(",")[1:] for l in ha[4:4 + 16384]] == [l.split(",")[1:] for l in hb[4:4 + 16384]]
    assert ha[4 + 16384:] == hb[4 + 16384:]
    da, db = oracle.db_read(a), oracle.db_read(b)
    assert da.contigs() == db.contigs() == ["chr1_test_contig_0", "chr2_test_contig_1", "chr3_test_contig_2"]
    kinds = set()
    for bi in range(0, 16384):
        xa, na = da.bin(bi)
        xb, nb = db.bin(bi)
        assert na == nb and np.array_equal(xa, xb), bi
        kinds.add(int(xa[0]))
    assert kinds == ({1} if enzyme == "cpf1" else {1, 2}) or kinds == {1}


@pytest.mark.gpu
def test_index_edge_cases_match_oracle(cli, oracle, tmp_path):
    """what the FASTA reader and the device site scan must get right: text before the first header is ignored, CRLF line
    ends, no final newline, soft-masked bases, contigs shorter than a site, an empty record, sites cut by N, a tandem
    repeat with more copies than Short.MaxValue (count and position list are capped at 32767, BlockReader.scala:147-153)"""
    rng = np.random.default_rng(12)
    unit = "ACGTTGCAAGCTTGACCATGAGG"                                        # one forward NGG site per copy
    body = "".join(rng.choice(list("ACGT"), 30000))
    recs = [("tiny", "ACGTACGTGG"), ("empty", ""), ("tandem repeat", unit * 33000), ("mixed\tcase", body[:9000].lower() + "N" + body[9000:20000] + "nnn" + body[20000:]),
            ("palin", "CC" + "ACGTACGTACGTACGTACGTA" + "GG")]                # a window that is a forward AND a reverse site
    fa = str(tmp_path / "edge.fa")
    with open(fa, "w", newline="") as f:
        f.write("ACGTACGTACGTACGTACGTAGG\r\n")                              # before any header: dropped (ReferenceEncoder.scala:116)
        for i, (name, s) in enumerate(recs):
            f.write(">%s\r\n" % name)
            for k in range(0, len(s), 61):
                last = i == len(recs) - 1 and k + 61 >= len(s)
                f.write(s[k:k + 61] + ("" if last else "\r\n"))
    a, b = str(tmp_path / "db_cli"), str(tmp_path / "db_oracle")
    subprocess.check_call([cli, "index", "--reference", fa, "--database", a, "--enzyme", "spcas9ngg"], stderr=subprocess.DEVNULL)
    assert oracle.lib.ffo_index_fasta(fa.encode(), b.encode(), b"spcas9ngg", 7) == 0, oracle.error()
    da, db = oracle.db_read(a), oracle.db_read(b)
    assert da.contigs() == db.contigs() == ["tiny", "empty", "tandem_repeat", "mixed_case", "palin"]
    capped = 0
    for bi in range(16384):
        xa, na = da.bin(bi)
        xb, nb = db.bin(bi)
        assert na == nb and np.array_equal(xa, xb), bi
        capped += int(np.sum((xa.view(np.uint64) >> np.uint64(48)) == 32767))
    assert capped >= 1


@pytest.mark.gpu
def test_index_gz_and_bgzf_is_plain_gzip(cli, tmp_path):
    """BGZF is a series of gzip members: Python's gzip module must read the body the CLI wrote"""
    import gzip
    fa = str(tmp_path / "g.fa")
    random_genome(fa, n_contigs=1, length=20000)
    subprocess.check_call(["gzip", "-k", fa])
    a, b = str(tmp_path / "db_plain"), str(tmp_path / "db_gz")
    subprocess.check_call([cli, "index", "-reference", fa, "-database", a], stderr=subprocess.DEVNULL)        # single-dash spelling
    subprocess.check_call([cli, "index", "--reference", fa + ".gz", "--database", b], stderr=subprocess.DEVNULL)
    assert gzip.open(a).read() == gzip.open(b).read()
    assert open(a + ".header").read() == open(b + ".header").read()
    raw = gzip.open(a).read()
    assert len(raw) % 8 == 0 and np.frombuffer(raw[:8], dtype="<i8")[0] in (1, 2)
    assert open(a, "rb").read()[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def test_cli_errors(cli, tmp_path):
    r = subprocess.run([cli, "index", "--reference", "/nonexistent.fa", "--database", str(tmp_path / "x")], capture_output=True)
    assert r.returncode == 1 and b"cannot open" in r.stderr
    r = subprocess.run([cli, "index", "--reference", "x", "--database", "y", "--enzyme", "cas13"], capture_output=True)
    assert r.returncode == 1 and b"Unable to find the correct parameter pack" in r.stderr
    r = subprocess.run([cli, "discover", "--fasta", "x"], capture_output=True)
    assert r.returncode == 1 and b"Missing required option" in r.stderr
    r = subprocess.run([cli, "frobnicate"], capture_output=True)
    assert r.returncode == 2


# ---- GPU box ------------------------------------------------------------------------------------------------------------
def table_rows(path):
    lines = open(path).read().split("\n")
    return lines[0], sorted(l for l in lines[1:] if l)


@pytest.mark.gpu
@pytest.mark.parametrize("enzyme,positions", [("spcas9ngg", True), ("spcas9ngg", False), ("cpf1", True)])
def test_discover_and_score_tables_are_byte_identical(cli, oracle, tmp_path, enzyme, positions):
    fa, gfa = str(tmp_path / "genome.fa"), str(tmp_path / "guides.fa")
    random_genome(fa, seed=7)
    if enzyme == "cpf1":
        seq = "".join(l.strip() for l in open(fa) if not l.startswith(">")).upper()
        with open(gfa, "w") as f:
            f.write(">myc_like\n" + seq[4800:6800] + "\n")  # spans the planted repeat
    else:
        guide_fasta(gfa, fa)
    db = str(tmp_path / "db")
    subprocess.check_call([cli, "index", "--reference", fa, "--database", db, "--enzyme", enzyme], stderr=subprocess.DEVNULL)
    out_cli, out_ora = str(tmp_path / "cli.sites"), str(tmp_path / "ora.sites")
    cmd = [cli, "discover", "--database", db, "--fasta", gfa, "--output", out_cli, "--maxMismatch", "4", "--maximumOffTargets", "2"]
    if positions:
        cmd.append("--positionOutput")
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    assert oracle.lib.ffo_discover_fasta(db.encode(), gfa.encode(), out_ora.encode(), 4, 2, 6, int(positions), 0, 0.0, 1.0) == 0, oracle.error()
    assert open(out_cli).read() == open(out_ora).read()
    assert "OVERFLOW" in open(out_cli).read() and "\tOK\t" in open(out_cli).read()
    # the same discover over three bin shards (three contexts on the one GPU) must not change a byte
    out_sh = str(tmp_path / "cli_sharded.sites")
    subprocess.check_call(cmd[:7] + [out_sh] + cmd[8:] + ["--devices", "0,0,0"], stderr=subprocess.DEVNULL)
    assert open(out_sh).read() == open(out_cli).read()
    if not positions:
        return  # a discover table without positions cannot be re-read once it carries per-hit scores (reference quirk 15)
    metrics = "hsu2013,doench2016cfd,minot,dangerous,JostAndSantos,reciprocalofftargets"
    for include, recip in ((False, 1), (True, 3)):
        s_cli, s_ora = str(tmp_path / "cli.scored"), str(tmp_path / "ora.scored")
        subprocess.check_call([cli, "score", "--input", out_cli, "--output", s_cli, "--scoringMetrics", metrics, "--database", db, "--maxReciprocalMismatch", str(recip)]
                              + (["--includeOTs"] if include else []), stderr=subprocess.DEVNULL)
        assert oracle.lib.ffo_score_file(db.encode(), out_ora.encode(), s_ora.encode(), metrics.encode(), 2 ** 31 - 1, int(include), recip) == 0, oracle.error()
        assert open(s_cli).read() == open(s_ora).read()
        col = open(s_cli).readline().rstrip("\n").split("\t").index("ReciprocalOffTargets")
        partners = [l.split("\t")[col] for l in open(s_cli).read().split("\n")[1:] if l]
        assert enzyme == "cpf1" or (any(x != "NA" for x in partners) and any(x == "NA" for x in partners))
    hdr = open(s_cli).readline().rstrip("\n").split("\t")
    if enzyme == "cpf1":  # CFD / Hsu2013 are dropped for non-Cas9 enzymes (ScoreResults.scala:111-118)
        assert "Hsu2013" not in hdr and "basesDiffToClosestHit" in hdr
        assert "JostCRISPRi_maxOT" not in hdr  # JostAndSantosCRISPRi.scala:53-58
    else:
        assert hdr[7:10] == ["Hsu2013", "DoenchCFD_maxOT", "DoenchCFD_specificityscore"] and "JostCRISPRi_specificityscore" in hdr
        assert "{Doench2016CFDScore=" in open(s_cli).read()


@pytest.mark.gpu
def test_empty_inputs_through_the_cli(cli, oracle, tmp_path):
    """a genome without a single target site and a guide file without a single guide: valid (empty) database, header-only tables"""
    fa, gfa = str(tmp_path / "empty.fa"), str(tmp_path / "guides.fa")
    with open(fa, "w") as f:
        f.write(">chrE\n" + "AT" * 500 + "\n")                      # no NGG / CCN anywhere
    with open(gfa, "w") as f:
        f.write(">g1\nGACTTGCATCCGAAGCCGGTGGG\n")
    db = str(tmp_path / "db")
    subprocess.check_call([cli, "index", "--reference", fa, "--database", db, "--enzyme", "spcas9ngg"], stderr=subprocess.DEVNULL)
    odb = oracle.db_read(db)
    assert odb.n_bins == 16384 and all(len(odb.bin(b)[0]) == 1 for b in range(0, 16384, 257))
    out = str(tmp_path / "out.sites")
    subprocess.check_call([cli, "discover", "--database", db, "--fasta", gfa, "--output", out], stderr=subprocess.DEVNULL)
    rows = open(out).read().split("\n")
    assert rows[0].startswith("contig\tstart") and rows[1].split("\t")[:4] == ["g1", "0", "23", "GACTTGCATCCGAAGCCGGTGGG"] and rows[1].split("\t")[-2:] == ["0", ""]
    with open(gfa, "w") as f:
        f.write(">nothing\nATATATATATATATATATATATATAT\n")          # no guide in here
    subprocess.check_call([cli, "discover", "--database", db, "--fasta", gfa, "--output", out], stderr=subprocess.DEVNULL)
    assert open(out).read().count("\n") == 1


@pytest.mark.gpu
def test_bulge_subcommand_writes_what_the_library_finds(cli, tmp_path):
    """`flashfry-hip bulge` (config C5: Cas12a, mismatches + one bulge; this repository's own table, the reference has no such
    search): its rows are exactly the library's hits for the guides the FASTA holds, guides in file order, hits in database order"""
    from flashfry_amd import capi
    fa = str(tmp_path / "genome.fa")
    random_genome(fa, n_contigs=2, length=40000, seed=12)
    db = str(tmp_path / "db")
    assert subprocess.run([cli, "index", "--reference", fa, "--database", db, "--enzyme", "cpf1"], capture_output=True).returncode == 0
    seq = "".join(l.strip() for l in open(fa) if not l.startswith(">")).upper()
    rng = np.random.default_rng(5)
    guides = []
    while len(guides) < 12:   # forward TTTN sites cut out of the genome; every other one loses a base (an RNA-bulge relative of its site)
        i = int(rng.integers(0, len(seq) - 30))
        w = seq[i:i + 24]
        if w.startswith("TTT") and set(w) <= set("ACGT") and not w.endswith("AAA"):
            if len(guides) % 2:
                k = int(rng.integers(6, 20))
                w = w[:k] + w[k + 1:] + seq[i + 24]
            guides.append(w)
    gf = str(tmp_path / "guides.fa")
    with open(gf, "w") as f:
        for k, w in enumerate(guides):
            f.write(">g%d\n%s\n" % (k, w))
    out = str(tmp_path / "bulge.tsv")
    r = subprocess.run([cli, "bulge", "--database", db, "--fasta", gf, "--output", out, "--maxMismatch", "2", "--maxBulge", "1"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rows = [l.rstrip("\n").split("\t") for l in open(out)]
    assert rows[0] == ["guide", "guideSequence", "offTarget", "count", "mismatches", "bulgeType", "bulgePosition"]
    rows = rows[1:]
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    enc = lambda w: sum(code[ch] << (2 * (23 - i)) for i, ch in enumerate(w))
    # the FASTA scan may find more than one site per record (both strands): take the guide sequences the CLI reports, in its order
    order = []
    for row in rows:
        if not order or order[-1] != (row[0], row[1]):
            order.append((row[0], row[1]))
    longs = np.array([enc(s) | (1 << 48) for _, s in order], dtype=np.uint64)
    with capi.Context(0) as ctx:
        ctx.open(db)
        res = ctx.discover_bulge(longs, 2, 1)
    want = []
    kinds = ["none", "RNA", "DNA"]
    dec = lambda v: "".join("ACGT"[(int(v) >> (2 * (23 - i))) & 3] for i in range(24))
    for gi, (name, s) in enumerate(order):
        for h in range(int(res.guide_offsets[gi]), int(res.guide_offsets[gi + 1])):
            t = int(res.hit_targets[h])
            want.append([name, s, dec(t), str(t >> 48), str(int(res.hit_mismatches[h])), kinds[int(res.hit_bulge_type[h])], str(int(res.hit_bulge_position[h]))])
    assert rows == want and len(rows) >= 12
    assert {r_[5] for r_ in rows} >= {"none", "RNA"}
    # a Cas9 database is refused
    db9 = str(tmp_path / "db9")
    assert subprocess.run([cli, "index", "--reference", fa, "--database", db9, "--enzyme", "spcas9ngg"], capture_output=True).returncode == 0
    r = subprocess.run([cli, "bulge", "--database", db9, "--fasta", gf, "--output", out], capture_output=True)
    assert r.returncode == 1 and b"Cas12a" in r.stderr
