"""The world frame of the trained / exported model (VERDICT r1 missing #3).

The reference's stages after `Train-Stage1` apply FIXED corrections to `splat.ply` --
  rotate_splat.py --rotations x:270,y:180,z:0      source/container/src/main.py:1481-1500
  mirror_splat.py --axis x                         main.py:1510-1523
-- which only make sense if the PLY is in the frame the upstream trainer exports: nerfstudio's (ColmapDataParser
axis swap = `applied_transform`, orientation "up" -> +z, centred on the cameras, unit cube) for `ns-train`, the
example Parser's normalisation for `simple_trainer.py`.  These tests pin that frame: a COLMAP model with a KNOWN
gravity direction goes in, and "up" has to come out on +z, then on +y after the reference's fixed rotations.
[UPSTREAM-UNVERIFIED: nerfstudio / gsplat are restated, not run.]
"""
import json
import math
import os

import numpy as np
import torch

from mi3dgs import colmap_json, dataset, io_colmap, io_ply, scenes, transform


def _ring_dataset(tmp, up, n_views=12, w=48, h=32, seed=3):
    """Cameras on a ring around the axis `up` through `target`, all looking at `target` with `up` as their up."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    up = np.asarray(up, dtype=np.float64) / np.linalg.norm(up)
    a = np.cross(up, [0.3, -0.5, 0.8]); a /= np.linalg.norm(a)
    b = np.cross(up, a)
    target = np.array([0.7, -1.1, 0.4])
    cams = [io_colmap.Camera(1, "PINHOLE", w, h, np.array([40.0, 40.0, w / 2, h / 2]))]
    imgs, vms = [], []
    os.makedirs(os.path.join(tmp, "images"), exist_ok=True)
    for i in range(n_views):
        th = 2 * math.pi * i / n_views
        eye = target + 3.0 * (math.cos(th) * a + math.sin(th) * b) + 1.2 * up
        V = scenes.look_at(torch.tensor(eye), torch.tensor(target), up=tuple(up)).double().numpy()
        vms.append(V)
        imgs.append(io_colmap.Image(i + 1, io_colmap.rotmat_to_qvec(V[:3, :3]), V[:3, 3].copy(), 1, f"img_{i:03d}.png"))
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(tmp, "images", f"img_{i:03d}.png"))
    # points: a flat blob in the plane perpendicular to `up` plus a "stick" along +up (the marker)
    flat = target + rng.normal(size=(400, 1)) * a * 1.0 + rng.normal(size=(400, 1)) * b * 0.5 + rng.normal(size=(400, 1)) * up * 0.02
    stick = target + np.linspace(0.0, 0.6, 60)[:, None] * up
    xyz = np.concatenate([flat, stick])
    rgb = rng.integers(0, 255, (len(xyz), 3), dtype=np.uint8)
    io_colmap.write_model(os.path.join(tmp, "sparse", "0"), cams, imgs, xyz, rgb)
    return np.stack(vms), xyz, up, target


def test_rotation_matrix_between():
    rng = np.random.default_rng(0)
    for _ in range(20):
        a, b = rng.normal(size=3), rng.normal(size=3)
        R = dataset.rotation_matrix_between(a, b)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-9) and abs(np.linalg.det(R) - 1) < 1e-9
        assert np.allclose(R @ (a / np.linalg.norm(a)), b / np.linalg.norm(b), atol=1e-9)
    z = np.array([0.0, 0.0, 1.0])
    assert np.allclose(dataset.rotation_matrix_between(z, z), np.eye(3))
    assert np.allclose(dataset.rotation_matrix_between(-z, z) @ -z, z, atol=1e-9)          # antiparallel: still a rotation


def test_nerfstudio_frame_is_z_up_centred_unit_cube(tmp_path):
    vms, xyz, up, target = _ring_dataset(str(tmp_path), up=[0.2, -0.9, 0.35])
    ds = dataset.load_colmap_dataset(str(tmp_path), 1, frame="nerfstudio", test_every=0)
    Q, s, t = ds.world_rot.double().numpy(), ds.scale, ds.world_shift.double().numpy()
    assert np.allclose(Q @ Q.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(Q) - 1) < 1e-6     # a rotation, no mirror
    assert np.allclose(Q @ up, [0, 0, 1], atol=1e-5)                                          # gravity axis -> +z
    centres = torch.linalg.inv(ds.viewmats.double())[:, :3, 3].numpy()
    assert np.abs(centres.mean(0)).max() < 1e-5                                               # center_method = "poses"
    assert abs(np.abs(centres).max() - 1.0) < 1e-5                                            # auto_scale_poses
    assert np.allclose(centres[:, 2], centres[0, 2], atol=1e-5)                               # the ring is horizontal
    # points follow the same similarity, and every camera still sees them at the same pixels
    assert np.allclose(ds.points.double().numpy(), s * xyz @ Q.T + t, atol=1e-5)
    p_w = torch.from_numpy(xyz[:80])
    for i in (0, 5):
        V0, Vn = torch.from_numpy(vms[i]), ds.viewmats[i].double()
        c0 = (V0[:3, :3] @ p_w.T).T + V0[:3, 3]
        cn = (Vn[:3, :3] @ ds.points[:80].double().T).T + Vn[:3, 3]
        assert torch.allclose(c0[:, :2] / c0[:, 2:], cn[:, :2] / cn[:, 2:], atol=1e-4)
        assert torch.allclose(cn[:, 2], s * c0[:, 2], atol=1e-5)                              # depths scale with the scene
    # the stick points up
    st = ds.points[-60:].double().numpy()
    d = st[-1] - st[0]
    assert np.allclose(d / np.linalg.norm(d), [0, 0, 1], atol=1e-4)


def test_nerfstudio_frame_agrees_with_transforms_json(tmp_path):
    """colmap_json.colmap_to_json (what the reference's colmap_to_nerfstudio_cam.py:52-63 calls) writes OpenGL
    camera-to-world matrices in the `applied_transform` world; the dataset's frame is that world re-oriented,
    centred and scaled: every pose must be the same similarity away."""
    _ring_dataset(str(tmp_path), up=[-0.6, 0.1, 0.8])
    colmap_json.colmap_to_json(tmp_path / "sparse" / "0", tmp_path, ply_filename="sparse_pc.ply")
    meta = json.load(open(tmp_path / "transforms.json"))
    assert np.allclose(np.array(meta["applied_transform"])[:, :3], dataset.APPLIED_TRANSFORM)
    ds = dataset.load_colmap_dataset(str(tmp_path), 1, frame="nerfstudio", test_every=0)
    P = dataset.APPLIED_TRANSFORM
    Rot = ds.world_rot.double().numpy() @ P.T                       # Q = Rot P
    frames = sorted(meta["frames"], key=lambda f: f["file_path"])
    o_json = np.array([np.array(f["transform_matrix"])[:3, 3] for f in frames])
    for i, f in enumerate(frames):
        M = np.array(f["transform_matrix"])
        c2w = torch.linalg.inv(ds.viewmats[i].double()).numpy()
        c2w_gl = c2w.copy()
        c2w_gl[:3, 1:3] *= -1                                        # our cameras are OpenCV, nerfstudio's OpenGL
        assert np.allclose(c2w_gl[:3, :3], Rot @ M[:3, :3], atol=1e-5)
        assert np.allclose(c2w_gl[:3, 3], ds.scale * Rot @ (M[:3, 3] - o_json.mean(0)), atol=1e-5)


def test_export_then_reference_rotations_lands_y_up(tmp_path):
    """splat.ply in the training frame -> rotate_splat x:270,y:180,z:0 -> mirror_splat x (the reference's argv,
    main.py:1481-1523): the scene's up axis ends on +y, what the reference's viewer-facing asset has."""
    _ring_dataset(str(tmp_path), up=[0.5, 0.5, -0.7])
    ds = dataset.load_colmap_dataset(str(tmp_path), 1, frame="nerfstudio", test_every=0)
    n = ds.points.shape[0]
    S = dict(means=ds.points.clone(), quats=torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1), scales=torch.full((n, 3), -4.0),
             opacities=torch.zeros(n), sh0=torch.zeros(n, 1, 3), shN=torch.zeros(n, 15, 3))
    ply = str(tmp_path / "splat.ply")
    io_ply.write_ply(ply, S)                                        # what ns-export leaves behind (training frame)
    assert transform.main_rotate(["-i", ply, "--rotations", "x:270,y:180,z:0"]) == 0
    assert transform.main_mirror(["-i", ply, "--axis", "x"]) == 0
    out = io_ply.read_ply(ply)["means"].double().numpy()
    stick = out[-60:]
    d = stick[-1] - stick[0]
    assert np.allclose(d / np.linalg.norm(d), [0, 1, 0], atol=1e-4)
    flat = out[:400]
    assert flat[:, 1].std() < 0.1 * min(flat[:, 0].std(), flat[:, 2].std())         # the ground plane is y = const


def test_gsplat_frame_normalisation(tmp_path):
    """simple_trainer.py path: similarity_from_cameras + PCA alignment as restated in dataset.gsplat_frame."""
    vms, xyz, up, target = _ring_dataset(str(tmp_path), up=[0.1, 0.7, 0.7])
    ds = dataset.load_colmap_dataset(str(tmp_path), 1, frame="gsplat", test_every=0)
    Q, s, t = ds.world_rot.double().numpy(), ds.scale, ds.world_shift.double().numpy()
    assert np.allclose(Q @ Q.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(Q) - 1) < 1e-6
    assert np.allclose(ds.points.double().numpy(), s * xyz @ Q.T + t, atol=1e-5)
    # the flat blob's normal (= the gravity axis) is the smallest principal axis -> z; its long side -> x
    pts = ds.points[:400].double().numpy()
    sd = pts.std(0)
    assert sd[0] > sd[1] > sd[2]
    assert abs(abs((Q @ up)[2]) - 1.0) < 2e-2
    # cameras: rotations stay orthonormal, pixels unchanged
    p_w = torch.from_numpy(xyz[:80])
    for i in (1, 7):
        V0, Vn = torch.from_numpy(vms[i]), ds.viewmats[i].double()
        assert torch.allclose(Vn[:3, :3] @ Vn[:3, :3].T, torch.eye(3, dtype=torch.float64), atol=1e-5)
        c0 = (V0[:3, :3] @ p_w.T).T + V0[:3, 3]
        cn = (Vn[:3, :3] @ ds.points[:80].double().T).T + Vn[:3, 3]
        assert torch.allclose(c0[:, :2] / c0[:, 2:], cn[:, :2] / cn[:, 2:], atol=1e-4)
    assert ds.scene_scale > 0.5
