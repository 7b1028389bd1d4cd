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


def _run(mode, nproc=2, extra=("--rehearse",), steps=6):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", str(steps), "--warmup", "2",
           "--scene", "lego", "--mode", mode, "--no-cpu-baseline", "--no-stage-profile", *extra]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
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
    # value = the timed steps plus their share of a refine pass; both of the reference's jobs are in the line
    pr = d["presets"][d["config"]["preset"]]
    assert abs(pr["it_per_s"] - d["value"]) < 1e-9 and pr["it_per_s_without_refine"] >= d["value"]
    assert abs(2 * 6 / d["value"] - (6 * pr["ms_per_step_without_refine"] + 0.06 * pr["refine"]["refine_ms"]) * 1e-3) < 1e-6


@pytest.mark.timeout(300)
def test_scene_shard_mode_two_ranks(dev):
    d = _run("scene-shard")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["mode"] == "scene-shard"
    # reduce-scatter + all-gather of 59 floats per Gaussian of capacity, (G - 1) / G of it crossing the links
    assert d["config"]["xgmi_bytes_per_rank_per_step"] == int(2 * 0.5 * 4 * 59 * 300_000)
    assert d["value"] > 0 and d["async_errors"] == 0


# ------------------------------------------------------------------ bench.py's own RCCL branch
@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["replicas", "scene-shard"])
def test_bench_runs_its_rccl_branch_at_world_one(dev, mode):
    """`bench.py --gpus N` creates its `nccl` process group, barriers with device_ids and reduces its clock over RCCL only when
    the driver launches it with N > 1 -- which no one-GPU box can.  `--force-dist` takes exactly those lines at world 1: a fresh
    child process under torch.distributed.run, the group created before anything else touches the GPU.  scene-shard additionally
    drives reduce_scatter_tensor / all_gather_into_tensor through the sharded optimiser and checks one step bit for bit against
    the dense Adam on the same gradients."""
    d = _run(mode, nproc=1, extra=("--force-dist", "--verify-shard-step", "--param-checksum"), steps=4)
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["process_group"] == "nccl" and d["async_errors"] == 0
    assert d["config"]["mode"] == mode and d["value"] > 0 and d["param_checksum"] is not None
    assert set(d["presets"]) == {"splatfacto", "simple_trainer"} and d["config"]["preset"] == "splatfacto"
    for pr in d["presets"].values():
        assert 0 < pr["it_per_s"] <= pr["it_per_s_without_refine"] and pr["refine"]["refine_ms"] > 0
    if mode == "scene-shard":
        assert d["shard_step_bit_exact"] is True and d["scaling"] == "strong"
    else:
        assert d["shard_step_bit_exact"] is None and d["scaling"] == "weak"
