"""Input-side kernels: exact k-NN, INTER_AREA downscale, uint8 image cache."""
import os

import numpy as np
import pytest
import torch

from helpers import small_scene
from mi3dgs import dataset, ops, trainer
from oracle import post_oracle as PO

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _clouds():
    g = torch.Generator().manual_seed(0)
    uni = torch.rand(5000, 3, generator=g)
    sfm = torch.cat([torch.randn(4000, 3, generator=g), torch.randn(1500, 3, generator=g) * 0.02 + 2.0,
                     torch.randn(40, 3, generator=g) * 300.0])                       # clusters + far outliers
    flat = torch.cat([torch.rand(3000, 2, generator=g), torch.zeros(3000, 1)], 1)    # a plane
    line = torch.cat([torch.rand(2000, 1, generator=g), torch.zeros(2000, 2)], 1) + 5.0
    dup = torch.rand(700, 3, generator=g).repeat(3, 1)                                # every point three times
    same = torch.ones(300, 3) * 0.25                                                 # zero extent
    offset = torch.rand(3000, 3, generator=g) * 1e-2 + 1000.0                        # tight cloud far from the origin
    shell = torch.nn.functional.normalize(torch.randn(8000, 3, generator=g), dim=1) * 3.0   # a surface: most cells empty
    return dict(uniform=uni, sfm=sfm, plane=flat, line=line, duplicates=dup, identical=same, offset=offset, shell=shell)


@pytest.mark.parametrize("name", list(_clouds()))
@pytest.mark.parametrize("k", [1, 3, 4])
def test_knn_equals_brute_force(name, k):
    pts = _clouds()[name]
    want = PO.knn_sq_dists(pts.numpy(), k)
    d2, idx = ops.knn(pts.to(DEV), k, want_idx=True)
    d2, idx = d2.cpu().double().numpy(), idx.cpu().long().numpy()
    assert np.allclose(d2, want, rtol=2e-6, atol=1e-12)
    assert np.all(np.diff(d2, axis=1) >= 0)
    # the reported neighbours are other points, distinct, and at the reported distances
    n = pts.shape[0]
    assert np.all(idx != np.arange(n)[:, None]) and np.all((idx >= 0) & (idx < n))
    if k > 1:
        assert np.all(np.sort(idx, 1)[:, 1:] != np.sort(idx, 1)[:, :-1])
    p = pts.double().numpy()
    assert np.allclose(((p[idx] - p[:, None, :]) ** 2).sum(-1), d2, rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 65])
def test_knn_tiny_inputs(n):
    pts = torch.rand(n, 3, generator=torch.Generator().manual_seed(n))
    d2, idx = ops.knn(pts.to(DEV), 3, want_idx=True)
    want = PO.knn_sq_dists(pts.numpy(), 3)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(d2.cpu().numpy()), fin) and np.array_equal(idx.cpu().numpy() >= 0, fin)
    assert np.allclose(d2.cpu().numpy()[fin], want[fin], rtol=2e-6)
    assert ops.knn(torch.empty(0, 3, device=DEV), 3).shape == (0, 3)
    with pytest.raises(Exception, match="k must be"):
        ops.knn(pts.to(DEV), 5)


def test_knn_large_cloud_against_sampled_brute_force_and_sfm_scales():
    g = torch.Generator().manual_seed(5)
    n = 400_000
    pts = torch.cat([torch.randn(n // 2, 3, generator=g) * torch.tensor([4.0, 4.0, 0.3]),
                     torch.rand(n // 2, 3, generator=g) * 2 - 1]).to(DEV)
    d2 = ops.knn(pts, 3)
    sel = torch.randperm(n, generator=g)[:2000].to(DEV)
    d = torch.cdist(pts[sel].double(), pts.double()).pow(2)
    d[torch.arange(2000, device=DEV), sel] = float("inf")
    want = torch.topk(d, 3, dim=1, largest=False).values
    assert torch.allclose(d2[sel].double(), want, rtol=2e-6, atol=1e-12)
    P = dataset.init_gaussians(pts, torch.zeros(n, 3, dtype=torch.uint8))
    assert torch.allclose(P["scales"][sel, 0].double(), torch.log(torch.sqrt(want.mean(1))), atol=1e-5)


@pytest.mark.parametrize("shape,out", [((36, 54, 3), (18, 27)), ((37, 53, 3), (18, 26)), ((1080, 1920, 3), (270, 480)),
                                       ((101, 67, 3), (33, 22)), ((16, 16, 1), (1, 1)), ((20, 30, 4), (20, 30)),
                                       ((9, 200, 3), (9, 66))])
def test_area_downscale_equals_oracle(shape, out):
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    t = torch.from_numpy(img).to(DEV)
    got_u8 = ops.image_downscale_area(t, *out).cpu().numpy().astype(np.int32)
    want_f = PO.area_downscale(img, *out, as_float=True)
    got_f = ops.image_downscale_area(t, *out, as_float=True).cpu().numpy()
    assert np.abs(got_f - want_f).max() < 2e-6
    # uint8: equal except where the exact mean sits on a rounding boundary to float precision
    want_u8 = PO.area_downscale(img, *out).astype(np.int32)
    bad = got_u8 != want_u8
    frac = np.abs((want_f * 255.0) % 1.0 - 0.5)
    assert np.all(frac[bad] < 1e-3) and np.abs(got_u8 - want_u8).max() <= 1
    with pytest.raises(Exception, match="larger"):
        ops.image_downscale_area(t, shape[0] + 1, shape[1])


def test_u8_image_cache_trains_like_the_float_targets():
    sc = small_scene(n=1500, seed=21, width=96, height=64, n_views=3, fx=90.0).to(DEV)
    imgs_u8 = (torch.rand(3, 64, 96, 3, generator=torch.Generator().manual_seed(1)) * 255).to(torch.uint8).to(DEV)
    for n in (1, 2, 3, 5, 7, 64 * 96 * 3):                                         # ragged tails of the 4-wide kernel
        flat = imgs_u8.reshape(-1)[:n].contiguous()
        assert torch.equal(ops.image_u8_to_f32(flat), flat.float() * (1.0 / 255.0))
    runs = []
    for images in (imgs_u8, imgs_u8.float() * (1.0 / 255.0)):
        tr = trainer.Trainer({k: v.clone() for k, v in sc.params.items()}, sc.viewmats, sc.Ks, images, 96, 64,
                             trainer.TrainConfig(densify=False, fuse_adam=False))
        losses = [tr.step(i % 3, want_loss=True) for i in range(6)]
        runs.append((losses, tr.model.p("means").clone()))
    # identical targets bit for bit (checked above); the steps differ only by float-atomic summation order
    assert np.allclose(runs[0][0], runs[1][0], rtol=1e-5) and torch.allclose(runs[0][1], runs[1][1], atol=1e-5)


def test_ensure_downscaled_images_and_device_resize_on_load(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    os.makedirs(tmp_path / "images")
    src = {}
    for i in range(3):
        src[f"f{i}.png"] = rng.integers(0, 256, size=(45, 70, 3), dtype=np.uint8)
        Image.fromarray(src[f"f{i}.png"]).save(tmp_path / "images" / f"f{i}.png")
    assert dataset.ensure_downscaled_images(str(tmp_path / "images"), "4") == 3
    for name, a in src.items():
        got = np.asarray(Image.open(tmp_path / "images_4" / name))
        assert got.shape == (11, 17, 3)                                            # max(1, int(h / k)), int(w / k)
        assert np.abs(got.astype(np.int32) - PO.area_downscale(a, 11, 17).astype(np.int32)).max() <= 1
    assert dataset.ensure_downscaled_images(str(tmp_path / "images"), "4") == 0     # already complete
    assert dataset.ensure_downscaled_images(str(tmp_path / "images"), "1") == 0
    assert dataset.ensure_downscaled_images(str(tmp_path / "images"), "abc") == 0


@pytest.mark.parametrize("u8", [True, False])
def test_resolution_schedule_trains_on_box_filtered_targets(u8):
    """splatfacto's coarse-to-fine schedule: level d renders (W//d, H//d) with K/d against the d x d block mean."""
    from oracle import gs_oracle as O
    W, H = 100, 72                                           # 72 // 4 = 18, 100 // 4 = 25; not divisible at d = 8
    sc = small_scene(n=1200, seed=31, width=W, height=H, n_views=2, fx=95.0).to(DEV)
    imgs_u8 = (torch.rand(2, H, W, 3, generator=torch.Generator().manual_seed(2)) * 255).to(torch.uint8).to(DEV)
    images = imgs_u8 if u8 else imgs_u8.float() / 255.0
    cfg = trainer.TrainConfig(densify=False, num_downscales=3, resolution_schedule=2, lr_means=0.0, lr_scales=0.0,
                              lr_quats=0.0, lr_opacities=0.0, lr_sh0=0.0, lr_shN=0.0, sh_degree_interval=1)
    tr = trainer.Trainer({k: v.clone() for k, v in sc.params.items()}, sc.viewmats, sc.Ks, images, W, H, cfg)
    seen = []
    for step in range(8):
        loss = tr.step(step % 2, want_loss=True)
        d = 2 ** max(3 - step // 2, 0)
        seen.append((tr.W, tr.H))
        assert (tr.W, tr.H) == (W // d, H // d)
        K = sc.Ks[step % 2].clone()
        K[:2] /= d
        P = sc.params
        sd = min(step, 3)
        img, _, _ = ops.rasterization(P["means"], P["quats"], P["scales"].exp(), torch.sigmoid(P["opacities"]),
                                      torch.cat([P["sh0"], P["shN"]], 1), sc.viewmats[step % 2][None], K[None], W // d,
                                      H // d, sh_degree=sd)
        src = imgs_u8[step % 2].cpu().numpy()[: (H // d) * d, : (W // d) * d]
        gt = torch.from_numpy(PO.area_downscale(src, H // d, W // d, as_float=True))[None]
        want = float(O.photometric_loss(img.double().cpu(), gt, 0.2))
        assert abs(loss - want) < 2e-5 * max(1.0, abs(want))
    assert seen[0] == (12, 9) and seen[-1] == (W, H)
    r, a = tr.render(sc.viewmats[0], sc.Ks[0])               # rendering is always full resolution
    assert r.shape == (1, H, W, 3)


UD_CASES = [("OPENCV", [0.09, -0.03, 0.0012, -0.0008], (290.0, 305.0, 163.0, 98.0)),
            ("FULL_OPENCV", [-0.1, 0.02, 0.0005, 0.0003, 0.001, 0.01, 0.002, 0.0], (300.0, 300.0, 160.0, 100.0)),
            ("SIMPLE_RADIAL", [-0.12], (280.0, 280.0, 158.0, 103.0)),
            ("OPENCV_FISHEYE", [0.04, -0.008, 0.001, 0.0], (150.0, 148.0, 161.0, 99.0))]


@pytest.mark.parametrize("model,tail,K", UD_CASES)
@pytest.mark.parametrize("ch", [3, 1])
def test_undistort_kernel_equals_oracle(model, tail, K, ch):
    from mi3dgs import undistort as ud
    w, h = 320, 200
    p = ud.make_plan(model, tail, K, (w, h))
    img = np.random.default_rng(len(tail) + ch).integers(0, 256, size=(h, w, ch), dtype=np.uint8)
    want = PO.undistort_image(img, p.k_src, p.k_dst, p.dist, p.out_size[1], p.out_size[0], p.fisheye)
    t = torch.from_numpy(img).to(DEV)
    got_f = ops.image_undistort(t, p.k_src, p.k_dst, p.dist, p.out_size[1], p.out_size[0], fisheye=p.fisheye, as_float=True)
    got_u = ops.image_undistort(t, p.k_src, p.k_dst, p.dist, p.out_size[1], p.out_size[0], fisheye=p.fisheye)
    assert got_f.shape == (p.out_size[1], p.out_size[0], ch)
    # float32 coordinates on a noise image: a 1e-4 px coordinate error times a 255-level step
    err = np.abs(got_f.cpu().numpy().astype(np.float64) * 255.0 - want)
    assert err.max() < 0.25 and err.mean() < 0.01
    assert np.abs(got_u.cpu().numpy().astype(np.float64) - want).max() < 0.76
    # zero outside the source: a camera looking far past the image border
    far = ops.image_undistort(t, p.k_src, (p.k_dst[0], p.k_dst[1], p.k_dst[2] - 5000.0, p.k_dst[3]), p.dist, 8, 8,
                              fisheye=p.fisheye)
    assert int(far.max()) == 0


def test_dataset_undistorts_on_load(tmp_path):
    """COLMAP OPENCV camera on disk -> pinhole targets: sizes, intrinsics and pixels follow the plan."""
    from PIL import Image
    from mi3dgs import io_colmap
    from mi3dgs import undistort as ud
    w, h = 240, 160
    K = np.array([210.0, 205.0, 121.0, 79.0])
    dist = np.array([-0.14, 0.03, 0.001, -0.002])
    cams = [io_colmap.Camera(1, "OPENCV", w, h, np.concatenate([K, dist]))]
    rng = np.random.default_rng(0)
    os.makedirs(tmp_path / "images")
    ims, pix = [], {}
    for i in range(3):
        name = f"v{i}.png"
        pix[name] = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        Image.fromarray(pix[name]).save(tmp_path / "images" / name)
        ims.append(io_colmap.Image(i + 1, np.array([1.0, 0, 0, 0]), np.array([0.1 * i, 0, 2.0]), 1, name))
    io_colmap.write_model(str(tmp_path / "sparse" / "0"), cams, ims, rng.normal(size=(50, 3)),
                          rng.integers(0, 255, (50, 3)).astype(np.uint8))
    ds = dataset.load_colmap_dataset(str(tmp_path), 1)
    plan = ud.make_plan("OPENCV", dist, K, (w, h))
    assert (ds.width, ds.height) == plan.out_size and all(p is not None for p in ds.undistort)
    assert np.allclose(ds.Ks[0].numpy(), [[plan.K_out[0], 0, plan.K_out[2]], [0, plan.K_out[1], plan.K_out[3]], [0, 0, 1]], atol=1e-3)
    got = ds.load_images([0, 2], DEV, as_u8=True)
    for j, name in enumerate(("v0.png", "v2.png")):
        want = PO.undistort_image(pix[name], plan.k_src, plan.k_dst, plan.dist, plan.out_size[1], plan.out_size[0])
        assert np.abs(got[j].cpu().numpy().astype(np.float64) - want).max() < 0.76
    f = ds.load_images([1], DEV)
    assert f.dtype == torch.float32 and f.shape == (1, plan.out_size[1], plan.out_size[0], 3) and float(f.max()) <= 1.0
    # opting out keeps the files as they are (and says so)
    raw = dataset.load_colmap_dataset(str(tmp_path), 1, undistort=False)
    assert (raw.width, raw.height) == (w, h) and torch.equal(raw.load_images([0], DEV, as_u8=True)[0].cpu(), torch.from_numpy(pix["v0.png"]))
