"""CPU tests of the callers and data formats either side of the hot path (SURVEY.md 8a rows
a1-a9, a14-a15 and 8f): COLMAP binary model, dataset normalisation, checkpoint / PLY / .splat
export, and the reference's exact command lines.  The PLY tests run the REFERENCE'S OWN
strict parser (oracle/_ref/splat_converter, built from /root/reference by oracle/Makefile)."""
import math
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from mi3dgs import cli, dataset, io_colmap, io_ply, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONVERTER = os.path.join(ROOT, "oracle", "_ref", "splat_converter")
WOLF = os.path.join(ROOT, "tests", "golden", "wolf_1k.ply")


def _fake_dataset(tmp, n_views=9, w=48, h=32, n_pts=300, seed=0):
    from PIL import Image
    sc = scenes.make_cube(n=n_pts, seed=seed, width=w, height=h, n_views=n_views, fx=40.0)
    cams = [io_colmap.Camera(1, "PINHOLE", w, h, np.array([40.0, 40.0, w / 2, h / 2]))]
    imgs = []
    os.makedirs(os.path.join(tmp, "images"), exist_ok=True)
    rng = np.random.default_rng(seed)
    for i in range(n_views):
        V = sc.viewmats[i].double().numpy()
        imgs.append(io_colmap.Image(i + 1, io_colmap.rotmat_to_qvec(V[:3, :3]), V[:3, 3].copy(), 1, f"img_{i:03d}.png"))
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(tmp, "images", f"img_{i:03d}.png"))
    xyz = sc.params["means"].double().numpy()
    rgb = rng.integers(0, 255, (n_pts, 3), dtype=np.uint8)
    io_colmap.write_model(os.path.join(tmp, "colmap", "sparse", "0"), cams, imgs, xyz, rgb)
    return sc, xyz, rgb


def test_colmap_binary_roundtrip(tmp_path):
    sc, xyz, rgb = _fake_dataset(str(tmp_path))
    sp = io_colmap.find_sparse_dir(str(tmp_path))
    assert sp.endswith(os.path.join("colmap", "sparse", "0"))
    cams = io_colmap.read_cameras(os.path.join(sp, "cameras.bin"))
    ims = io_colmap.read_images(os.path.join(sp, "images.bin"))
    x2, c2, err = io_colmap.read_points3d(os.path.join(sp, "points3D.bin"))
    assert cams[1].model == "PINHOLE" and cams[1].pinhole() == (40.0, 40.0, 24.0, 16.0)
    assert len(ims) == 9 and np.array_equal(x2, xyz) and np.array_equal(c2, rgb)
    for i in range(9):
        assert np.allclose(ims[i + 1].world_to_camera(), sc.viewmats[i].double().numpy(), atol=1e-6)
    # the reference's own quality gate reads the count the same way (main.py:406-417)
    assert io_colmap.read_points3d_count(os.path.join(sp, "points3D.bin")) == 300
    assert struct.unpack("<Q", open(os.path.join(sp, "points3D.bin"), "rb").read(8))[0] == 300
    assert io_colmap.read_points3d_count("/nonexistent/points3D.bin") == 0


def test_colmap_layout_before_the_move_is_found_too(tmp_path):
    _fake_dataset(str(tmp_path))
    shutil.move(os.path.join(tmp_path, "colmap", "sparse"), os.path.join(tmp_path, "sparse"))   # gsplat Parser layout
    assert io_colmap.find_sparse_dir(str(tmp_path)).endswith(os.path.join("sparse", "0"))
    with pytest.raises(FileNotFoundError):
        io_colmap.find_sparse_dir(str(tmp_path / "images"))


def test_dataset_normalisation_and_split(tmp_path):
    sc, xyz, rgb = _fake_dataset(str(tmp_path))
    ds = dataset.load_colmap_dataset(str(tmp_path), 1)
    assert ds.width == 48 and ds.height == 32 and len(ds.image_paths) == 9
    assert ds.eval_idx == [0, 8] and len(ds.train_idx) == 7            # every 8th image held out
    centres = torch.linalg.inv(ds.viewmats)[:, :3, 3]
    assert abs(float(centres.abs().max()) - 1.0) < 1e-5                 # farthest camera at distance 1 (max-abs)
    assert float(centres.mean(0).abs().max()) < 1e-5
    # projection is invariant under the similarity: same pixels before and after
    p_w = torch.from_numpy(xyz[:50]).float()
    for i in (0, 4):
        V0, Vn = sc.viewmats[i], ds.viewmats[i]
        c0 = (V0[:3, :3] @ p_w.T).T + V0[:3, 3]
        cn = (Vn[:3, :3] @ ds.points[:50].T).T + Vn[:3, 3]
        assert torch.allclose(c0[:, :2] / c0[:, 2:], cn[:, :2] / cn[:, 2:], atol=1e-4)
    m, s = ds.denormalise(ds.points, torch.zeros(300, 3))
    assert torch.allclose(m, torch.from_numpy(xyz).float(), atol=1e-4) and abs(float(s[0, 0]) + math.log(ds.scale)) < 1e-6
    imgs = ds.load_images([0, 1], "cpu")
    assert imgs.shape == (2, 32, 48, 3) and 0 <= float(imgs.min()) and float(imgs.max()) <= 1


def test_downscaled_image_dir_and_intrinsics(tmp_path):
    from PIL import Image
    _fake_dataset(str(tmp_path))
    os.makedirs(tmp_path / "images_2")
    for f in os.listdir(tmp_path / "images"):
        Image.open(tmp_path / "images" / f).resize((24, 16)).save(tmp_path / "images_2" / f)
    ds = dataset.load_colmap_dataset(str(tmp_path), 2)
    assert (ds.width, ds.height) == (24, 16) and ds.image_paths[0].endswith(os.path.join("images_2", "img_000.png"))
    assert torch.allclose(ds.Ks[0], torch.tensor([[20.0, 0, 12.0], [0, 20.0, 8.0], [0, 0, 1]]))


def test_sfm_initialisation():
    g = torch.Generator().manual_seed(0)
    pts = torch.rand(500, 3, generator=g)
    rgb = torch.randint(0, 255, (500, 3), generator=g, dtype=torch.uint8)
    # the k-NN itself is a HIP kernel (tests/test_gpu_input_side.py); here the oracle's distances go in
    from oracle import post_oracle as PO
    P = dataset.init_gaussians(pts, rgb, knn_d2=torch.from_numpy(PO.knn_sq_dists(pts.numpy(), 3).mean(1)).float())
    with pytest.raises(ValueError, match="GPU"):
        dataset.init_gaussians(pts, rgb)                   # no CPU path for the search
    d = torch.cdist(pts.double(), pts.double())
    ref = torch.log(torch.sqrt((torch.topk(d, 4, largest=False).values[:, 1:] ** 2).mean(1)))
    assert torch.allclose(P["scales"][:, 0].double(), ref, atol=1e-4) and torch.equal(P["scales"][:, 0], P["scales"][:, 2])
    assert torch.allclose(torch.sigmoid(P["opacities"]), torch.full((500,), 0.1), atol=1e-6)
    assert torch.allclose(0.5 + dataset.SH_C0 * P["sh0"][:, 0], rgb.float() / 255, atol=1e-6)
    assert float(P["shN"].abs().max()) == 0 and P["shN"].shape == (500, 15, 3)


def _random_splats(n=257, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(means=torch.randn(n, 3, generator=g), quats=torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1),
                scales=torch.randn(n, 3, generator=g) - 3, opacities=torch.randn(n, generator=g) * 2,
                sh0=torch.randn(n, 1, 3, generator=g) * 0.5, shN=torch.randn(n, 15, 3, generator=g) * 0.1)


def test_ply_schema_and_roundtrip(tmp_path):
    S = _random_splats()
    p = str(tmp_path / "splat.ply")
    assert io_ply.write_ply(p, S) == 257
    raw = open(p, "rb").read()
    lines = raw[: raw.index(b"end_header\n")].decode().splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and lines[2] == "element vertex 257"
    assert [l.split()[2] for l in lines[3:]] == io_ply.PLY_FIELDS and all(l.startswith("property float ") for l in lines[3:])
    assert len(raw) == raw.index(b"end_header\n") + 11 + 257 * 62 * 4
    R = io_ply.read_ply(p)
    for k in S:
        assert torch.equal(R[k], S[k]), k
    # f_rest is channel-major, exactly as the reference's converter script lays it out
    # (post_processing/gsplat_pt_to_ply.py:64: shN.transpose(1, 2).flatten(start_dim=1))
    rows = io_ply.splats_to_rows(S)
    assert torch.equal(rows[:, 9:54], S["shN"].transpose(1, 2).flatten(start_dim=1))
    assert torch.equal(rows[:, 6:9], S["sh0"].transpose(1, 2).flatten(start_dim=1))
    assert torch.equal(rows[:, 3:6], torch.zeros(257, 3)) and torch.equal(rows[:, 58:62], S["quats"])


def test_nonfinite_gaussians_are_dropped(tmp_path):
    S = _random_splats(10)
    S["means"][3, 1] = float("nan")
    S["scales"][7, 0] = float("inf")
    assert io_ply.write_ply(str(tmp_path / "a.ply"), S) == 8


def test_reference_written_ply_is_read_and_rewritten_byte_for_byte(tmp_path):
    """tests/golden/wolf_1k.ply was written by the reference's own saveSplatToPly."""
    R = io_ply.read_ply(WOLF)
    assert R["means"].shape == (1000, 3) and all(not torch.isnan(v).any() for v in R.values())
    # the reference's SPZ decoder writes opacity = +inf for alpha 255/255: data, not corruption
    assert int(torch.isinf(R["opacities"]).sum()) == 13 and torch.isfinite(R["means"]).all()
    out = str(tmp_path / "wolf_copy.ply")
    assert io_ply.write_ply(out, R, drop_nonfinite=False) == 1000
    assert open(out, "rb").read() == open(WOLF, "rb").read()
    assert io_ply.write_ply(out, R) == 987


@pytest.mark.skipif(not os.path.isfile(CONVERTER), reason="oracle/_ref/splat_converter not built")
def test_ply_is_accepted_by_the_reference_spz_codec(tmp_path):
    """our PLY -> reference loadSplatFromPly -> SPZ -> reference loadSpz -> PLY: within the
    codec's quantisation (24-bit fixed-point xyz, u8 log-scale / alpha / colour, 8-bit quats)."""
    S = _random_splats(500, seed=3)
    S["means"] = S["means"].clamp(-3, 3)
    p = str(tmp_path / "m.ply")
    io_ply.write_ply(p, S)
    r = subprocess.run([CONVERTER, p], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.isfile(str(tmp_path / "m.spz")), r.stderr
    os.rename(p, str(tmp_path / "orig.ply"))
    r = subprocess.run([CONVERTER, str(tmp_path / "m.spz")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    R = io_ply.read_ply(p)
    assert R["means"].shape == (500, 3)
    assert float((R["means"] - S["means"]).abs().max()) < 2e-3
    assert float((R["scales"] - S["scales"]).abs().max()) < 0.04            # 1/16 log-units per step
    assert float((torch.sigmoid(R["opacities"]) - torch.sigmoid(S["opacities"])).abs().max()) < 1 / 255 + 1e-3
    assert float(((0.5 + 0.2820948 * R["sh0"]) - (0.5 + 0.2820948 * S["sh0"])).abs().max()) < 0.03
    qa, qb = torch.nn.functional.normalize(R["quats"], dim=1), S["quats"]
    qerr = 1 - (qa * qb).sum(1).abs()          # three 8-bit components, w reconstructed
    assert float(qerr.max()) < 8e-3 and float(qerr.mean()) < 5e-4
    assert float((R["shN"] - S["shN"]).abs().max()) < 0.07                  # 4-5 bit SH quantisation


def test_checkpoint_schema_is_what_the_reference_exporter_loads(tmp_path):
    """post_processing/gsplat_pt_to_ply.py:38-73: sorted(listdir)[-1], torch.load(weights_only=True)["splats"]."""
    S = _random_splats(64)
    ck = tmp_path / "ckpts"
    io_ply.save_checkpoint(str(ck / "ckpt_6999_rank0.pt"), S, 6999)
    io_ply.save_checkpoint(str(ck / "ckpt_14999_rank0.pt"), S, 14999)
    last = sorted(os.listdir(ck))[-1]
    c = torch.load(str(ck / last), map_location=torch.device("cpu"), weights_only=True)
    sp = c["splats"]
    assert set(sp) == {"means", "sh0", "shN", "opacities", "scales", "quats"}
    assert sp["means"].shape == (64, 3) and sp["sh0"].shape == (64, 1, 3) and sp["shN"].shape == (64, 15, 3)
    assert sp["opacities"].shape == (64,) and sp["scales"].shape == (64, 3) and sp["quats"].shape == (64, 4)
    # the exact tensor expressions of the reference script give our PLY rows
    f_dc = sp["sh0"].detach().transpose(1, 2).flatten(start_dim=1)
    f_rest = sp["shN"].detach().transpose(1, 2).flatten(start_dim=1)
    rows = io_ply.splats_to_rows(S)
    assert torch.equal(rows[:, 6:9], f_dc) and torch.equal(rows[:, 9:54], f_rest) and torch.equal(rows[:, 54], sp["opacities"])


def test_splat_file_layout(tmp_path):
    S = _random_splats(100)
    p = str(tmp_path / "a.splat")
    assert io_ply.write_splat(p, S) == 100 and os.path.getsize(p) == 3200
    rec = np.fromfile(p, dtype=[("pos", "<f4", 3), ("scale", "<f4", 3), ("rgba", "u1", 4), ("rot", "u1", 4)])
    key = rec["scale"].prod(1) * rec["rgba"][:, 3] / 255.0
    assert (np.diff(key) <= 1e-4 * np.abs(key[:-1]).max()).all()           # sorted by importance


def test_reference_command_lines_parse(tmp_path):
    # source/container/src/main.py:1270-1306 (+ the image-cache flags inserted at :2126-2145)
    argv = ["splatfacto", "--timestamp", "train-stage-1", "--viewer.quit-on-train-completion=True",
            "--logging.local-writer.enable", "False", "--logging.profiler", "none",
            "--pipeline.model.use_scale_regularization=True", "--max-num-iterations", "15000",
            "--pipeline.datamanager.cache-images", "cpu", "--pipeline.datamanager.max-thread-workers", "1",
            "colmap", "--data", "/opt/ml/input/data/training/dataset", "--downscale-factor", "2"]
    a = cli.parse_ns_train(argv)
    assert a["model"] == "splatfacto" and a["dataparser"] == "colmap" and a["data"].endswith("dataset")
    assert a["downscale"] == 2 and a["max_steps"] == 15000 and a["timestamp"] == "train-stage-1" and a["scale_reg"]
    cfg = cli.splatfacto_config("splatfacto", 15000, True, 100, 1000)
    assert cfg.use_scale_regularization and cfg.prune_opa == 0.1 and cfg.grow_grad2d == 0.0008 and cfg.absgrad
    assert cli.splatfacto_config("splatfacto-big", 1, False, 1, 1).prune_opa == 0.005
    # main.py:1328-1338
    b = cli.parse_simple_trainer(["default", "--max_steps", "15000", "--result-dir", "/d/exports", "--data_factor", "1",
                                  "--steps_scaler", "0.25", "--disable_viewer", "--packed", "--batch-size", "1",
                                  "--data-dir", "/d"])
    assert b["strategy"] == "default" and b["max_steps"] == 15000 and b["steps_scaler"] == 0.25 and b["data_dir"] == "/d"
    c = cli.simple_trainer_config(b, 1000)
    assert c.max_steps == 3750 and c.refine_every == 25 and c.reset_every == 750 and c.sh_degree_interval == 250
    with pytest.raises(SystemExit):
        cli.parse_ns_train(["nerfacto"]) and cli.main_ns_train(["nerfacto", "colmap", "--data", "x"])
    with pytest.raises(SystemExit):
        cli.parse_simple_trainer(["bogus"])


def test_shims_exist_and_are_executable():
    base = os.path.join(ROOT, "pipeline-pointcloud_amd", "shims")
    for rel in ("ns-train", "ns-export", os.path.join("gsplat", "examples", "simple_trainer.py")):
        assert os.access(os.path.join(base, rel), os.X_OK), rel


def test_ns_export_shim_writes_splat_ply(tmp_path):
    S = _random_splats(32)
    ck = str(tmp_path / "nerfstudio_models" / "step-000000009.ckpt")
    io_ply.save_checkpoint(ck, S, 9)
    cfgp = tmp_path / "config.yml"
    cfgp.write_text("# mi3dgs\n" + '{"engine": "mi3dgs", "checkpoint": "%s"}\n' % ck)
    assert cli.main_ns_export(["gaussian-splat", "--load-config", str(cfgp), "--output-dir", str(tmp_path / "exports")]) == 0
    R = io_ply.read_ply(str(tmp_path / "exports" / "splat.ply"))
    assert torch.equal(R["means"], S["means"])


def test_colmap_to_json_shim_writes_transforms_and_point_cloud(tmp_path, capsys):
    """`training/colmap_to_nerfstudio_cam.py -d D` (reference main.py:1220-1226): transforms.json in nerfstudio's
    conventions and the ASCII point cloud beside the sparse model."""
    import json
    from mi3dgs import colmap_json
    rng = np.random.default_rng(4)
    cams = [io_colmap.Camera(1, "OPENCV", 640, 480, np.array([500.0, 505.0, 321.0, 239.0, -0.1, 0.02, 0.001, -0.002]))]
    ims = []
    for i in range(5):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        ims.append(io_colmap.Image(i + 1, q, rng.normal(size=3), 1, f"frame_{i:03d}.jpg"))
    xyz = rng.normal(size=(40, 3)) * 3.0
    rgb = rng.integers(0, 256, (40, 3)).astype(np.uint8)
    io_colmap.write_model(str(tmp_path / "sparse" / "0"), cams, ims, xyz, rgb)
    assert colmap_json.main(["-d", str(tmp_path)]) == 0
    assert "creating transforms.json file" in capsys.readouterr().out
    T = json.load(open(tmp_path / "transforms.json"))
    assert (T["w"], T["h"], T["camera_model"]) == (640, 480, "OPENCV")
    assert [T[k] for k in ("fl_x", "fl_y", "cx", "cy", "k1", "k2", "p1", "p2")] == [500.0, 505.0, 321.0, 239.0, -0.1, 0.02, 0.001, -0.002]
    assert T["ply_file_path"] == f"{tmp_path}/sparse/0/sparse.ply" and len(T["frames"]) == 5
    A = np.array(T["applied_transform"])
    assert A.tolist() == [[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0]]
    for fr, im in zip(T["frames"], ims):
        assert fr["file_path"] == f"images/{im.name}" and fr["colmap_im_id"] == im.id
        c2w = np.array(fr["transform_matrix"])
        assert np.allclose(c2w[3], [0, 0, 0, 1])
        # a world point seen through the json pose (OpenGL axes, transformed world) = the COLMAP camera point with y, z negated
        X = rng.normal(size=3)
        x_cv = (im.world_to_camera() @ np.append(X, 1.0))[:3]
        x_gl = (np.linalg.inv(c2w) @ np.append(A[:, :3] @ X + A[:, 3], 1.0))[:3]
        assert np.allclose(x_gl, x_cv * [1, -1, -1], atol=1e-9)
    lines = open(tmp_path / "sparse" / "0" / "sparse.ply").read().splitlines()
    assert lines[:3] == ["ply", "format ascii 1.0", "element vertex 40"] and lines[9] == "end_header" and len(lines) == 50
    row = lines[10].split()
    assert np.allclose([float(v) for v in row[:3]], A[:, :3] @ xyz[0], atol=1e-5) and [int(v) for v in row[3:]] == rgb[0].tolist()
    # the reference script's soft failures: messages, exit code 0
    assert colmap_json.main(["-d", str(tmp_path / "nope")]) == 0 and "doesn't exist" in capsys.readouterr().out
    os.makedirs(tmp_path / "empty")
    assert colmap_json.main(["-d", str(tmp_path / "empty")]) == 0 and "Sparse path does not currently exist" in capsys.readouterr().out
    # two cameras are refused the way nerfstudio refuses them
    io_colmap.write_model(str(tmp_path / "two" / "sparse" / "0"), cams + [io_colmap.Camera(2, "PINHOLE", 640, 480, np.array([500.0, 500, 320, 240]))],
                          ims, xyz, rgb)
    with pytest.raises(RuntimeError, match="Only single camera"):
        colmap_json.main(["-d", str(tmp_path / "two")])


def test_colmap_reader_against_bytes_laid_out_from_the_published_format(tmp_path):
    """the reader had only ever read files written by this repository's own writer.  Here the three
    files are laid out byte by byte from COLMAP's documented binary format (src/colmap/scene/reconstruction_io.cc:
    little-endian; cameras: u64 count, then per camera i32 id, i32 model, u64 width, u64 height, f64 params[];
    images: u64 count, then per image i32 id, f64 qvec[4], f64 tvec[3], i32 camera_id, NUL-terminated name, u64 n_points2D,
    n x (f64 x, f64 y, i64 point3D_id); points3D: u64 count, then per point u64 id, f64 xyz[3], u8 rgb[3], f64 error,
    u64 track_length, track x (i32 image_id, i32 point2D_idx)) -- with several camera models, observations and tracks of
    different lengths, i.e. records of different sizes that a reader has to step over correctly."""
    import struct
    from mi3dgs import io_colmap
    d = tmp_path / "sparse" / "0"
    d.mkdir(parents=True)
    cams = [(7, 1, 1920, 1080, [1450.0, 1440.0, 960.0, 540.0]),                     # PINHOLE
            (3, 2, 800, 600, [700.0, 400.0, 300.0, -0.05]),                         # SIMPLE_RADIAL
            (11, 4, 640, 480, [500.0, 501.0, 320.0, 240.0, 0.1, -0.02, 1e-3, -2e-3])]   # OPENCV
    b = struct.pack("<Q", len(cams))
    for cid, model, w, h, p in cams:
        b += struct.pack("<iiQQ", cid, model, w, h) + struct.pack(f"<{len(p)}d", *p)
    (d / "cameras.bin").write_bytes(b)
    imgs = [(5, (0.5, 0.5, -0.5, 0.5), (0.1, -0.2, 3.0), 7, "a/frame_0005.png", [(10.5, 20.25, 42), (1.0, 2.0, -1)]),
            (2, (1.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 3, "b.jpg", []),
            (9, (0.0, 0.0, 1.0, 0.0), (-1.5, 2.5, 0.5), 11, "c with space.JPG", [(0.0, 0.0, 7)] * 5)]
    b = struct.pack("<Q", len(imgs))
    for iid, q, t, cid, name, obs in imgs:
        b += struct.pack("<i4d3di", iid, *q, *t, cid) + name.encode() + b"\x00" + struct.pack("<Q", len(obs))
        for x, y, pid in obs:
            b += struct.pack("<ddq", x, y, pid)
    (d / "images.bin").write_bytes(b)
    pts = [(42, (1.0, 2.0, 3.0), (255, 0, 7), 0.5, [(5, 0), (9, 3)]),
           (7, (-4.5, 0.25, 1e-3), (1, 2, 3), 1.25, [(9, 0)]),
           (100000000000, (0.0, 0.0, 0.0), (128, 128, 128), 0.0, [])]
    b = struct.pack("<Q", len(pts))
    for pid, xyz, rgb, err, track in pts:
        b += struct.pack("<Q3d3BdQ", pid, *xyz, *rgb, err, len(track))
        for iid, k in track:
            b += struct.pack("<ii", iid, k)
    (d / "points3D.bin").write_bytes(b)

    assert io_colmap.read_points3d_count(str(d / "points3D.bin")) == 3          # main.py:406-417
    C = io_colmap.read_cameras(str(d / "cameras.bin"))
    assert sorted(C) == [3, 7, 11]
    assert (C[7].model, C[7].width, C[7].height) == ("PINHOLE", 1920, 1080) and C[7].pinhole() == (1450.0, 1440.0, 960.0, 540.0)
    assert C[3].model == "SIMPLE_RADIAL" and C[3].pinhole() == (700.0, 700.0, 400.0, 300.0) and C[3].distortion().tolist() == [-0.05]
    assert C[11].model == "OPENCV" and C[11].distortion().tolist() == [0.1, -0.02, 1e-3, -2e-3]
    I = io_colmap.read_images(str(d / "images.bin"))
    assert sorted(I) == [2, 5, 9]
    assert I[5].name == "a/frame_0005.png" and I[5].camera_id == 7 and I[5].qvec.tolist() == [0.5, 0.5, -0.5, 0.5]
    assert I[5].tvec.tolist() == [0.1, -0.2, 3.0]
    assert I[2].name == "b.jpg" and I[9].name == "c with space.JPG" and I[9].camera_id == 11 and I[9].tvec.tolist() == [-1.5, 2.5, 0.5]
    W = I[5].world_to_camera()
    import numpy as np
    assert np.allclose(W[:3, :3] @ W[:3, :3].T, np.eye(3), atol=1e-12) and np.allclose(W[:3, 3], [0.1, -0.2, 3.0])
    xyz, rgb, err = io_colmap.read_points3d(str(d / "points3D.bin"))
    assert xyz.tolist() == [[1.0, 2.0, 3.0], [-4.5, 0.25, 1e-3], [0.0, 0.0, 0.0]]
    assert rgb.tolist() == [[255, 0, 7], [1, 2, 3], [128, 128, 128]] and err.tolist() == [0.5, 1.25, 0.0]
