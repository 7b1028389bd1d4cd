"""BASELINE.json configs[3] (independent scenes, one per GPU) and configs[4] (one scene, view-per-rank with the sharded
optimiser) through `bench.py` itself, as a 2-rank rehearsal on the one GPU of the test box at S1 size: gloo carries the
collectives (`--rehearse`), every rank uses cuda:0.  On a node the same command line runs one rank per GPU over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--rehearse", "--scene", "lego", "--mode", mode, "--no-cpu-baseline", "--no-stage-profile"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-500:]           # rank 0 prints ONE json line
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_replica_mode_two_ranks(dev):
    d = _run("replicas")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6 and d["warmup"] == 2
    assert d["config"]["mode"] == "replicas" and d["config"]["gaussians"] == 300_000 and d["config"]["xgmi_bytes_per_rank_per_step"] == 0
    assert d["value"] > 0 and abs(d["value"] - 2 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]      # whole-job aggregate
    assert d["async_errors"] == 0 and d["dtype"] == "f32" and d["vs_baseline"] is None


@pytest.mark.timeout(300)
def test_scene_shard_mode_two_ranks(dev):
    d = _run("scene-shard")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["mode"] == "scene-shard"
    # reduce-scatter + all-gather of 59 floats per Gaussian of capacity, (G - 1) / G of it crossing the links
    assert d["config"]["xgmi_bytes_per_rank_per_step"] == int(2 * 0.5 * 4 * 59 * 300_000)
    assert d["value"] > 0 and d["async_errors"] == 0
