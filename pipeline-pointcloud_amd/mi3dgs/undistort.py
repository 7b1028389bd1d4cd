"""Lens undistortion plan for COLMAP cameras: what nerfstudio's datamanager computes with OpenCV
before splatfacto trains (`_undistort_image`: getOptimalNewCameraMatrix(alpha=0) + undistort + crop to
the valid ROI for perspective cameras, fisheye.estimateNewCameraMatrixForUndistortRectify(balance=0)
for fisheye ones), reached through source/container/src/main.py:1303-1306.  [UPSTREAM-UNVERIFIED]:
neither nerfstudio nor cv2 is in this image; the published OpenCV algorithms are restated (float64,
host side: a few hundred flops per camera) and the per-pixel work runs in
`mi3dgs_image_undistort` (csrc/spatial.hip).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

PERSPECTIVE = ("SIMPLE_RADIAL", "RADIAL", "OPENCV", "FULL_OPENCV")
FISHEYE = ("OPENCV_FISHEYE", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE")


@dataclass
class Plan:
    fisheye: bool
    dist: List[float]                   # OpenCV order
    k_src: Tuple[float, float, float, float]   # distorted camera, OpenCV pixel convention (cx - 0.5)
    k_dst: Tuple[float, float, float, float]   # pinhole camera of the CROPPED output, same convention
    src_size: Tuple[int, int]           # (w, h) the source image has to have
    out_size: Tuple[int, int]           # (w, h) of the undistorted, cropped image
    K_out: Tuple[float, float, float, float]   # intrinsics to train with (cx + 0.5 restored)


def opencv_coefficients(model: str, dist: Sequence[float]) -> Optional[Tuple[bool, List[float]]]:
    """COLMAP model + its distortion tail -> (fisheye?, OpenCV coefficient vector); None = pinhole."""
    d = [float(x) for x in dist]
    if model in ("SIMPLE_PINHOLE", "PINHOLE") or not any(abs(x) > 0 for x in d):
        return None
    if model == "SIMPLE_RADIAL":
        return False, [d[0], 0.0, 0.0, 0.0]
    if model == "RADIAL":
        return False, [d[0], d[1], 0.0, 0.0]
    if model == "OPENCV":
        return False, [d[0], d[1], d[2], d[3]]
    if model == "FULL_OPENCV":
        return False, [d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]]
    if model == "OPENCV_FISHEYE":
        return True, [d[0], d[1], d[2], d[3]]
    if model == "SIMPLE_RADIAL_FISHEYE":
        return True, [d[0], 0.0, 0.0, 0.0]
    if model == "RADIAL_FISHEYE":
        return True, [d[0], d[1], 0.0, 0.0]
    raise ValueError(f"camera model {model} is not supported")


def distort_normalised(x, y, dist, fisheye):
    """Forward lens model on normalised coordinates (numpy, any shape)."""
    d = list(dist) + [0.0] * (8 - len(dist))
    if not fisheye:
        r2 = x * x + y * y
        rad = (1 + r2 * (d[0] + r2 * (d[1] + r2 * d[4]))) / (1 + r2 * (d[5] + r2 * (d[6] + r2 * d[7])))
        return (x * rad + 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x),
                y * rad + d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y)
    r = np.sqrt(x * x + y * y)
    th = np.arctan(r)
    t2 = th * th
    thd = th * (1 + t2 * (d[0] + t2 * (d[1] + t2 * (d[2] + t2 * d[3]))))
    sc = np.where(r > 1e-8, thd / np.maximum(r, 1e-300), 1.0)
    return x * sc, y * sc


def undistort_points(px, py, K, dist, fisheye, iters=None):
    """Pixel coordinates of the distorted image -> normalised undistorted coordinates (cv::undistortPoints:
    5 fixed-point iterations for the perspective model; Newton on theta for the fisheye one)."""
    fx, fy, cx, cy = K
    x0, y0 = (np.asarray(px, np.float64) - cx) / fx, (np.asarray(py, np.float64) - cy) / fy
    d = list(dist) + [0.0] * (8 - len(dist))
    if not fisheye:
        x, y = x0.copy(), y0.copy()
        for _ in range(5 if iters is None else iters):
            r2 = x * x + y * y
            icd = (1 + r2 * (d[5] + r2 * (d[6] + r2 * d[7]))) / (1 + r2 * (d[0] + r2 * (d[1] + r2 * d[4])))
            dx = 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x)
            dy = d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
            x, y = (x0 - dx) * icd, (y0 - dy) * icd
        return x, y
    thd = np.sqrt(x0 * x0 + y0 * y0)
    thd = np.clip(thd, -np.pi / 2, np.pi / 2)
    th = thd.copy()
    for _ in range(10 if iters is None else iters):
        t2 = th * th
        f = th * (1 + t2 * (d[0] + t2 * (d[1] + t2 * (d[2] + t2 * d[3])))) - thd
        fp = 1 + t2 * (3 * d[0] + t2 * (5 * d[1] + t2 * (7 * d[2] + t2 * 9 * d[3])))
        th = th - f / fp
    sc = np.where(thd > 1e-8, np.tan(th) / np.maximum(thd, 1e-300), 1.0)
    return x0 * sc, y0 * sc


def _rectangles(K, dist, size, newK=None, n=9):
    """cv icvGetRectangles: inscribed rectangle of the undistorted 9x9 boundary grid."""
    w, h = size
    gx, gy = np.meshgrid(np.arange(n) * (w - 1) / (n - 1), np.arange(n) * (h - 1) / (n - 1))
    x, y = undistort_points(gx, gy, K, dist, False)
    if newK is not None:
        x, y = x * newK[0] + newK[2], y * newK[1] + newK[3]
    ix0, ix1 = x[:, 0].max(), x[:, -1].min()
    iy0, iy1 = y[0, :].max(), y[-1, :].min()
    return ix0, iy0, ix1 - ix0, iy1 - iy0


def optimal_new_camera(K, dist, size):
    """cv::getOptimalNewCameraMatrix(K, dist, size, alpha=0): (newK, roi=(x, y, w, h))."""
    w, h = size
    ix, iy, iw, ih = _rectangles(K, dist, size)
    fx0, fy0 = (w - 1) / iw, (h - 1) / ih
    newK = (fx0, fy0, -fx0 * ix, -fy0 * iy)
    rx, ry, rw, rh = _rectangles(K, dist, size, newK)
    x0, y0 = int(np.ceil(rx)), int(np.ceil(ry))
    x1, y1 = int(np.floor(rx + rw)), int(np.floor(ry + rh))
    x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
    return newK, (x0, y0, max(x1 - x0, 0), max(y1 - y0, 0))


def fisheye_new_camera(K, dist, size):
    """cv::fisheye::estimateNewCameraMatrixForUndistortRectify(K, D, size, I, balance=0)."""
    w, h = size
    fx, fy, cx, cy = K
    px = np.array([w / 2, w, w / 2, 0.0])
    py = np.array([0.0, h / 2, h, h / 2])
    x, y = undistort_points(px, py, K, dist, True)
    aspect = fx / fy
    cnx, cny = x.mean(), y.mean() * aspect
    y = y * aspect
    f1, f2 = w * 0.5 / (cnx - x.min()), w * 0.5 / (x.max() - cnx)
    f3, f4 = h * 0.5 * aspect / (cny - y.min()), h * 0.5 * aspect / (y.max() - cny)
    f = max(f1, f2, f3, f4)                                  # balance = 0
    return (f, f / aspect, -cnx * f + w * 0.5, (-cny * f + h * aspect * 0.5) / aspect)


def make_plan(model: str, dist_tail: Sequence[float], K_colmap, size) -> Optional[Plan]:
    """K_colmap = (fx, fy, cx, cy) as stored (already scaled to `size` = (w, h)).  None = nothing to do."""
    oc = opencv_coefficients(model, dist_tail)
    if oc is None:
        return None
    fisheye, dist = oc
    w, h = int(size[0]), int(size[1])
    fx, fy, cx, cy = (float(v) for v in K_colmap)
    k_src = (fx, fy, cx - 0.5, cy - 0.5)                     # OpenCV wants the pixel centre at integers
    if fisheye:
        nk = fisheye_new_camera(k_src, dist, (w, h))
        roi = (0, 0, w, h)
        ow, oh = w, h
    else:
        nk, roi = optimal_new_camera(k_src, dist, (w, h))
        ow, oh = min(roi[2] + 1, w - roi[0]), min(roi[3] + 1, h - roi[1])      # image[y : y + h + 1, x : x + w + 1]
    if ow < 1 or oh < 1:
        raise ValueError("undistortion leaves no valid pixels")
    k_dst = (nk[0], nk[1], nk[2] - roi[0], nk[3] - roi[1])
    return Plan(fisheye, list(dist), k_src, k_dst, (w, h), (ow, oh), (k_dst[0], k_dst[1], k_dst[2] + 0.5, k_dst[3] + 0.5))
