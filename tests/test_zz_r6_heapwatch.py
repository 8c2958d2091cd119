"""Regression test of round 6's root cause (profiles/r06/uaf_analysis.md): with the checker in the library's process and EVERY freed heap
chunk of the process poisoned, parked and verified (tools/heapwatch.c, LD_PRELOAD), a slice of the randomised parity sweep on a busy box
must leave no chunk damaged.  Rounds 1-5 destroyed their HIP streams in ffh_destroy; hipStreamDestroy -> amd::HostQueue::terminate()
deletes the queue's roc::VirtualGPU under the runtime's own signal-handler thread, whose late stores then land in whatever took the
920-byte chunk next (twice in ~500 000 cases: the checker's int[229] arrays).  The library now pools its streams (ffh_streams.hpp)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _heapwatch_so():
    so = os.path.join(ROOT, "tools", "libheapwatch.so")
    src = os.path.join(ROOT, "tools", "heapwatch.c")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-o", so, src, "-ldl", "-lpthread"])
    return so


def _sweep(seconds, workers, extra_env, tmp_path, tag):
    so = _heapwatch_so()
    procs = []
    for k in range(workers):
        env = dict(os.environ, LD_PRELOAD=so, HEAPWATCH_LOG=str(tmp_path / ("hw_%s_%d" % (tag, k))), HEAPWATCH_SEGV="1", HEAPWATCH_BT="900-940", FFH_POOL_DEBUG="1", **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), str(seconds), str(91000 + 100 * k), "--oracle", "inproc", "--quiet", "--seal"],
                                      cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    cases, damaged, outs = 0, 0, []
    for p in procs:
        out, _ = p.communicate(timeout=seconds + 240)
        outs.append(out)
        m = re.search(r"all (\d+) cases agree \(inproc oracle, pool errors 0, heapwatch (\d+) damaged chunks", out)
        assert p.returncode == 0 and m, out[-3000:]
        cases += int(m.group(1))
        damaged += int(m.group(2))
    return cases, damaged, outs


def test_no_write_after_free_in_the_process(tmp_path):
    """eight in-process sweep workers side by side (the runtime's handler thread is late on a busy box: that is when rounds 1-5's
    hipStreamDestroy let it write into freed memory), 40 seconds: parity everywhere, not one freed chunk of any of the processes written to"""
    cases, damaged, outs = _sweep(40, 8, {}, tmp_path, "pooled")
    assert cases >= 200, cases
    assert damaged == 0, "\n".join(o[-1500:] for o in outs if "WRITE AFTER FREE" in o or "HEAPWATCH" in o)
