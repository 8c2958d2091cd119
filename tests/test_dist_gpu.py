"""Two ranks with real HIP shards on the test box's GPU: the sharded step of bench.py -- HIP scan per shard +
device-resident exchange over torch.distributed -- must equal the unsharded discover, with the ordered cut-off crossing the shard
boundary.  (gloo moves the device tensors here; one rank per GPU over RCCL runs the same code on a multi-GPU node.)"""
import json
import os
import subprocess
import sys

import pytest

from tests.test_dist_cpu import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,max_ot", [(2, 40), (2, 2000), (3, 15)])
def test_hip_shards_plus_exchange_equal_the_unsharded_discover(tmp_path, world, max_ot):
    out = str(tmp_path / "res.json")
    env = dict(os.environ, OMP_NUM_THREADS="1", FFH_TEST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker_gpu.py"), out, "4", str(max_ot)]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.load(open(out))
    assert res["world"] == world
    for k in ("ok_hits", "ok_ints", "ok_max", "ok_two_pass"):
        assert res[k], (k, res)
    assert res["max_sum_err"] <= 1e-9, res
    if max_ot < 2000:   # the cut-off really crossed a shard boundary, and some guides reached it only in a later shard
        assert 0 < res["n_overflowed"] < res["n_guides"] and res["crossing"] > 0 and res["cut_in_later_shard"] > 0, res


@pytest.mark.parametrize("world", [2, 3])
def test_cas12a_bulge_search_over_bin_shards_equals_the_unsharded_search(tmp_path, world):
    """config C5's multi-GPU half: every rank searches its bin shard (HIP, seeded bulge search), the hit lists concatenated in rank
    order are the unsharded search's (no cut-off, no score: nothing to exchange on the data path)"""
    out = str(tmp_path / "res.json")
    env = dict(os.environ, OMP_NUM_THREADS="1", FFH_TEST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker_bulge.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    res = json.load(open(out))
    assert res["world"] == world and res["ok"], res
    assert res["n_hits"] > 1000 and res["types"] == [0, 1, 2], res
