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
