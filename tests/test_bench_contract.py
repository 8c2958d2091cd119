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


def check_contract(d, n_gpus, scaling=None):   # (N = 1: "scaling" is null -- nothing scales)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == scaling
    assert d["unit"] == "comparisons/s" and d["dtype"] == "u64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # the HBM figure of SURVEY.md section 8d stays in achieved / peak / frac; `bound` names what binds the kernel
    assert r["bound"] in ("hbm", "valu-issue") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
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
    # the bench checks its own step: aggregates-only summaries == list-delivering discover, sampled hit lists == brute force
    assert d["verified"] is True and d["discover_with_lists_ms"] > 0
    assert d["executed_pair_tests_per_s"] > 0 and d["roofline"]["useful_valu_frac"] > 0
    assert d["skewed"] and "error" not in d["skewed"] and d["skewed"]["raw_hits"] > 0
    assert d["real_genome"] is None  # no FF_GENOME_FASTA on the box
    c2 = d["c2"]                     # BASELINE.json configs[1]: the chr22-scale step and the CLI's wall time (BGZF database file -> table)
    assert c2 and "error" not in c2 and "cli_error" not in c2 and c2["ms_per_step"] > 0 and c2["discover_wall_warm_s"] > 0 and c2["table_bytes"] > 1000, c2


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_two_rank_line_on_one_gpu(scaling):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FFH_BENCH_SAME_GPU="1", FFH_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", scaling] + SMALL + ["--cpu-seconds", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    check_contract(d, 2, scaling)
    assert len(d["per_rank"]) == 2 and [r["rank"] for r in d["per_rank"]] == [0, 1] and all(r["compare_ms"] > 0 for r in d["per_rank"])
    assert d["config"]["targets_total"] > d["config"]["targets_per_gpu"] and d["config"]["parallelism"] == "bin-shard x2"
    if scaling == "strong":  # ONE database split by bins: the ranks' shards add up to the single-GPU database (3e6 drawn, duplicates collapse)
        assert 2.9e6 < d["config"]["targets_total"] < 3.1e6
    assert d["hits"]["kept_positions"] > 0


def test_sharded_step_with_one_rank_goes_through_the_library_exchange():
    """FFH_BENCH_FORCE_EXCHANGE=1: the sharded step of `bench.py --gpus N` with a single rank -- ffh_comm_create_rank (ncclCommInitRank,
    world 1) + ffh_discover_sharded, the unique id handed round by torch.distributed -- the code the driver's multi-GPU run executes"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FFH_BENCH_FORCE_EXCHANGE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--cpu-seconds", "0", "--no-skewed", "--no-c2", "--no-verify"], env=env,
                       capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    check_contract(d, 1)
    assert "RCCL collectives issued by libflashfry_hip (rccl-rank)" in d["config"]["exchange"], d["config"]
    assert d["hits"]["kept_positions"] > 0
