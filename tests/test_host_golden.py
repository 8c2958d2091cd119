"""The reference's golden fixtures through the PRODUCT's host layer (flashfry_amd/host/*.cpp) -- not through the oracle (VERDICT r4
weak 2 / next 3 iii).  tests/host_golden_main.cpp is a test-only main over ffhost_table.cpp / ffhost_core.cpp; it links the HIP library
(the host layer's other half) but makes no GPU call, so these run under -m "not gpu"."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN = {1: 24, 2: 23, 3: 23, 4: 23, 5: 22, 6: 22}


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


@pytest.fixture(scope="module")
def host_golden(tmp_path_factory):
    from flashfry_amd import _build
    _build.build_hip_library()
    exe = str(tmp_path_factory.mktemp("host_golden") / "host_golden")
    host = os.path.join(ROOT, "flashfry_amd", "host")
    # (round 6: the host layer under AddressSanitizer + UBSan for these runs -- it is what a drop-in CLI user's tables go through; leaks are
    # not looked for: the HIP runtime the library links keeps what it allocates at load time)
    os.environ.setdefault("ASAN_OPTIONS", "detect_leaks=0")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-Wall", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "host_golden_main.cpp"),
                           os.path.join(host, "ffhost_core.cpp"), os.path.join(host, "ffhost_table.cpp"), os.path.join(host, "ffhost_index.cpp"),
                           "-L" + _build.LIB_DIR, "-lflashfry_hip", "-Wl,-rpath," + _build.LIB_DIR, "-lz", "-lpthread"])
    return exe


@pytest.fixture(scope="module")
def ka(golden_dir):
    with open(os.path.join(golden_dir, "known_answers.json")) as f:
        return json.load(f)


def test_fake_sites_read_then_written_is_byte_identical(host_golden, golden_dir, tmp_path):
    """TabDelimitedHanderTest.scala:40-51 "read the same file it just wrote": TabDelimitedInput -> TabDelimitedOutput with no score
    models, compared by content (the reference compares MD5s)"""
    src, out = os.path.join(golden_dir, "fake.sites"), str(tmp_path / "fake.sites_temp")
    msg = subprocess.run([host_golden, "roundtrip", src, out], capture_output=True, text=True, check=True).stdout
    assert msg.strip() == "99 guides"
    assert open(out, "rb").read() == open(src, "rb").read()


def test_site_finder_known_answers(host_golden, ka, tmp_path):
    """SimpleSiteFinderTest.scala:13-173 through ffhost::findTargetSites (what `discover` uses to find the guides of a FASTA)"""
    for k, (enzyme, flank, seq, expected, src) in enumerate(ka["site_cases"]):
        fa = tmp_path / ("case%d.fa" % k)
        fa.write_text(">ctg\n" + seq + "\n")
        rows = [ln.split(" ") for ln in subprocess.run([host_golden, "sites", str(enzyme), str(flank), str(fa)], capture_output=True, text=True, check=True).stdout.splitlines()]
        L = SCAN[enzyme]
        assert len(rows) == len(expected), src
        for (bases, start, fwd, has_ctx, ctx), (eb, es, ef, ec) in zip(rows, expected):
            assert (bases, int(start), fwd == "1", has_ctx == "1") == (eb, es, ef, ec), src
            if ec and flank:
                window = seq[es - flank:es + L + flank]
                assert ctx == (window if ef else revcomp(window)), src


def test_encoding_known_answers(host_golden, ka):
    """BitEncodingTest.scala: round trip and the mismatch literals through ffhost::BitEncoding"""
    s, c, _ = ka["roundtrip_case"]
    v, back, cnt = subprocess.run([host_golden, "encode", "2", s, str(c)], capture_output=True, text=True, check=True).stdout.split()
    assert (back, int(cnt)) == (s, c) and int(v) >> 48 == c
    for enz, s1, c1, s2, c2, exp, src in ka["mismatch_cases"]:
        got = subprocess.run([host_golden, "mismatches", str(enz), s1, str(c1), s2, str(c2)], capture_output=True, text=True, check=True).stdout
        assert int(got) == exp, src
    for bad in (["encode", "2", "ACGTN", "1"], ["encode", "2", "A" * 25, "1"], ["encode", "2", "ACGT", "0"]):   # BitEncoding.scala:47-60
        assert subprocess.run([host_golden] + bad, capture_output=True, text=True).returncode == 1
