"""bench.py prints ONE JSON line with the contract's keys -- single GPU, and the sharded step with two ranks (sharing the one
GPU of the test box over gloo; on a multi-GPU node the same code runs one rank per GPU over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SMALL = ["--targets", "3e6", "--guides", "3000", "--steps", "2", "--warmup", "1", "--no-traffic"]


def last_json(stdout):
    lines = [l for l in stdout.decode().splitlines() if l.strip()]
    assert lines and lines[-1].startswith("{"), lines[-3:]
    return json.loads(lines[-1])


def check_contract(d, n_gpus):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "comparisons/s" and d["dtype"] == "u64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - d["config"]["guides"] * d["config"]["targets_total"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--cpu-seconds", "1"], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    check_contract(d, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["hits"]["raw"] > 0


def test_two_rank_line_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FFH_BENCH_SAME_GPU="1", FFH_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL + ["--cpu-seconds", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    check_contract(d, 2)
    assert d["config"]["targets_total"] > d["config"]["targets_per_gpu"] and d["config"]["parallelism"] == "bin-shard x2"
    assert d["hits"]["kept_positions"] > 0
