"""Pins of the input-side oracle (k-NN, INTER_AREA) against independent implementations present
in this image: scikit-learn (what splatfacto itself calls) and PIL's BOX filter."""
import numpy as np
from PIL import Image

from oracle import post_oracle as PO


def test_knn_oracle_equals_sklearn():
    from sklearn.neighbors import NearestNeighbors
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.normal(size=(1500, 3)), rng.normal(size=(500, 3)) * 0.01 + 3.0]).astype(np.float32)
    d, _ = NearestNeighbors(n_neighbors=4, algorithm="auto", metric="euclidean").fit(pts).kneighbors(pts)
    ours = PO.knn_sq_dists(pts, 3)
    assert np.allclose(np.sqrt(ours), d[:, 1:], rtol=1e-6, atol=1e-9)


def test_knn_oracle_small_and_duplicate_inputs():
    assert PO.knn_sq_dists(np.zeros((1, 3)), 3).tolist() == [[np.inf] * 3]
    d = PO.knn_sq_dists(np.array([[0, 0, 0], [1, 0, 0], [1, 0, 0]], dtype=np.float32), 3)
    assert d[1].tolist() == [0.0, 1.0, np.inf] and d[0].tolist() == [1.0, 1.0, np.inf]


def test_area_oracle_against_pil_box_and_closed_forms():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(36, 54, 3), dtype=np.uint8)
    for (h, w) in ((18, 27), (36, 54), (12, 18), (1, 1), (9, 54), (4, 6)):          # integer factors
        ours = PO.area_downscale(img, h, w).astype(np.int32)
        pil = np.asarray(Image.fromarray(img).resize((w, h), Image.BOX), dtype=np.int32)
        assert np.abs(ours - pil).max() <= 1, (h, w)
    blk = img.reshape(18, 2, 18, 3, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.allclose(PO.area_downscale(img, 18, 18, as_float=True) * 255.0, blk)
    assert np.array_equal(PO.area_downscale(img, 36, 54), img)
    # fractional factors: a ramp p[i] = i has the closed-form window mean (F(b) - F(a)) / (b - a),
    # F(t) = integral of floor(x) over [0, t)
    W, w = 53, 17
    ramp = np.tile(np.arange(W, dtype=np.uint8)[None, :, None], (5, 1, 1))
    F = lambda t: np.floor(t) * (np.floor(t) - 1) / 2 + np.floor(t) * (t - np.floor(t))      # noqa: E731
    a, b = np.arange(w) * W / w, np.minimum((np.arange(w) + 1) * W / w, W)
    want = (F(b) - F(a)) / (b - a)
    got = PO.area_downscale(ramp, 2, w, as_float=True)[0, :, 0] * 255.0
    assert np.allclose(got, want, atol=1e-9)
    assert np.array_equal(PO.area_downscale(np.full((7, 9, 3), 200, np.uint8), 3, 4), np.full((3, 4, 3), 200, np.uint8))


# ------------------------------------------------------------------ lens undistortion (host-side plan)
import pytest  # noqa: E402

from mi3dgs import undistort as ud  # noqa: E402

CASES = [("SIMPLE_RADIAL", [-0.08], (820.0, 820.0, 470.0, 275.0)),
         ("RADIAL", [-0.11, 0.02], (800.0, 800.0, 480.0, 270.0)),
         ("OPENCV", [0.09, -0.03, 0.0012, -0.0008], (790.0, 805.0, 483.0, 268.0)),
         ("FULL_OPENCV", [-0.1, 0.02, 0.0005, 0.0003, 0.001, 0.01, 0.002, 0.0], (800.0, 800.0, 480.0, 270.0)),
         ("OPENCV_FISHEYE", [0.04, -0.008, 0.001, 0.0], (420.0, 418.0, 478.0, 272.0)),
         ("RADIAL_FISHEYE", [0.03, 0.004], (430.0, 430.0, 480.0, 270.0))]


@pytest.mark.parametrize("model,tail,K", CASES)
def test_undistort_points_inverts_the_lens_model(model, tail, K):
    fisheye, dist = ud.opencv_coefficients(model, tail)
    px, py = np.meshgrid(np.linspace(0, 959, 9), np.linspace(0, 539, 7))
    x, y = ud.undistort_points(px, py, K, dist, fisheye, iters=40)
    xd, yd = ud.distort_normalised(x, y, dist, fisheye)
    assert np.abs(xd * K[0] + K[2] - px).max() < 1e-6 and np.abs(yd * K[1] + K[3] - py).max() < 1e-6


@pytest.mark.parametrize("model,tail,K", CASES)
def test_plan_keeps_every_output_pixel_inside_the_source_image(model, tail, K):
    """alpha = 0 / balance = 0: the undistorted, cropped image has no invalid border."""
    w, h = 960, 540
    p = ud.make_plan(model, tail, K, (w, h))
    assert p is not None and p.src_size == (w, h) and 0.8 * w <= p.out_size[0] <= w and 0.8 * h <= p.out_size[1] <= h
    u, v = np.meshgrid(np.arange(p.out_size[0], dtype=np.float64), np.arange(p.out_size[1], dtype=np.float64))
    x, y = (u - p.k_dst[2]) / p.k_dst[0], (v - p.k_dst[3]) / p.k_dst[1]
    xd, yd = ud.distort_normalised(x, y, p.dist, p.fisheye)
    us, vs = p.k_src[0] * xd + p.k_src[2], p.k_src[1] * yd + p.k_src[3]
    slack = 1.0 if not p.fisheye else 0.02 * w       # fisheye: OpenCV's estimate uses the 4 edge midpoints only
    assert us.min() > -slack and us.max() < w - 1 + slack and vs.min() > -slack and vs.max() < h - 1 + slack
    # the principal points carry nerfstudio's half-pixel shift back and forth
    assert abs(p.K_out[2] - p.k_dst[2] - 0.5) < 1e-12 and abs(p.k_src[2] - (K[2] - 0.5)) < 1e-12
    # the optical axis stays where it was: normalised (0, 0) maps to the source principal point
    assert abs(p.k_src[0] * 0 + p.k_src[2] - (K[2] - 0.5)) < 1e-12


def test_pinhole_and_zero_distortion_need_no_plan():
    assert ud.make_plan("PINHOLE", [], (800, 800, 480, 270), (960, 540)) is None
    assert ud.make_plan("SIMPLE_RADIAL", [0.0], (800, 800, 480, 270), (960, 540)) is None
    with pytest.raises(ValueError, match="not supported"):
        ud.make_plan("THIN_PRISM_FISHEYE", [0.1] * 8, (800, 800, 480, 270), (960, 540))


def test_undistort_oracle_recovers_a_pinhole_picture_of_a_smooth_scene():
    """Independent of any implementation detail: a smooth function of the viewing ray, photographed
    through the lens model and undistorted, equals the same function photographed by the pinhole camera."""
    w, h = 320, 200
    K = (260.0, 255.0, 161.0, 98.0)
    p = ud.make_plan("OPENCV", [-0.15, 0.04, 0.002, -0.001], K, (w, h))

    def scene(x, y):
        return 127.5 + 100.0 * np.sin(3.0 * x + 0.5) * np.cos(2.0 * y - 0.3)

    ud_px, ud_py = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    xn, yn = ud.undistort_points(ud_px, ud_py, p.k_src, p.dist, False, iters=40)      # ray of each distorted pixel
    photo = np.clip(np.rint(scene(xn, yn)), 0, 255).astype(np.uint8)[..., None]
    got = PO.undistort_image(photo, p.k_src, p.k_dst, p.dist, p.out_size[1], p.out_size[0])[..., 0]
    u, v = np.meshgrid(np.arange(p.out_size[0], dtype=np.float64), np.arange(p.out_size[1], dtype=np.float64))
    want = scene((u - p.k_dst[2]) / p.k_dst[0], (v - p.k_dst[3]) / p.k_dst[1])
    assert np.abs(got - want)[2:-2, 2:-2].max() < 1.5          # quantisation + bilinear error of a smooth image
