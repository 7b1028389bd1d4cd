"""Oracle for the reference's PLY post-processing.  TEST INFRASTRUCTURE ONLY.

numpy / scipy restatement of
  source/container/src/pipeline/post_processing/rotate_splat.py:90-178   rotate_gaussians,
      rotate_sh_coefficients, :245-276 create_rotation_matrix, :278-310 parse_rotation_spec,
      :351-356 sequential application
  source/container/src/pipeline/post_processing/mirror_splat.py:33-124   mirror_ply
The reference modules themselves import `plyfile`, which is absent from this image, so they
cannot be imported here; their array arithmetic is restated 1:1 (scipy is present, so the very
same `Rotation` calls are used).  PINNED by nothing but the reference source: it ships no
fixtures for these scripts.
"""
import numpy as np
from scipy.spatial.transform import Rotation


def create_rotation_matrix(axis, angle_degrees):
    a = np.radians(angle_degrees)
    c, s = np.cos(a), np.sin(a)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def parse_rotation_spec(spec):
    out = []
    for part in (spec or "").split(","):
        if ":" in part:
            axis, angle = part.split(":")
            axis = axis.strip().lower()
            if axis in ("x", "y", "z"):
                try:
                    out.append((axis, float(angle.strip())))
                except ValueError:
                    pass
    return out


def rotate_sh_coefficients(sh_dc, sh_rest, R):
    """rotate_splat.py:141-178: only f_rest[:, 0:9], three at a time, times R^T."""
    rest = sh_rest.copy()
    if rest.shape[1] >= 9:
        for i in range(0, 9, 3):
            rest[:, i:i + 3] = np.dot(sh_rest[:, i:i + 3], R.T)
    return sh_dc.copy(), rest


def rotate_gaussians(positions, rotations_wxyz, sh_dc, sh_rest, R):
    """rotate_splat.py:90-139 (the per-Gaussian scipy loop, vectorised: same composition)."""
    pos = np.dot(positions, R.T)
    rot = Rotation.from_matrix(R)
    existing = Rotation.from_quat(np.column_stack([rotations_wxyz[:, 1], rotations_wxyz[:, 2], rotations_wxyz[:, 3],
                                                   rotations_wxyz[:, 0]]))
    q = (rot * existing).as_quat()
    quat = np.column_stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]])
    dc, rest = rotate_sh_coefficients(sh_dc, sh_rest, R)
    return pos, quat, dc, rest


def mirror(positions, rotations_wxyz, sh_rest, axis):
    """mirror_splat.py:52-124."""
    M = np.eye(3)
    M["xyz".index(axis), "xyz".index(axis)] = -1
    pos = np.dot(positions, M)
    rm = Rotation.from_quat(np.column_stack([rotations_wxyz[:, 1], rotations_wxyz[:, 2], rotations_wxyz[:, 3],
                                             rotations_wxyz[:, 0]])).as_matrix()
    mr = np.einsum("ij,njk->nik", M, rm)
    mask = np.linalg.det(mr) < 0.0
    mr[mask, :, 0] *= -1.0
    q = Rotation.from_matrix(mr).as_quat()
    quat = np.column_stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]])
    rest = sh_rest.copy()
    if rest.shape[1] >= 9:
        for i in range(0, 9, 3):
            rest[:, i:i + 3] = np.dot(rest[:, i:i + 3], M)
    return pos, quat, rest


# ------------------------------------------------------------------ input side
def knn_sq_dists(points, k=3):
    """Squared distances to the k nearest OTHER points, ascending, float64 brute force; rows with
    fewer than k other points are padded with +inf.  What splatfacto's `k_nearest_sklearn`
    (NearestNeighbors(k+1), first column dropped) and gsplat's `knn(points, 4)[:, 1:]` return,
    squared; pinned against scikit-learn itself in tests/test_input_side_cpu.py."""
    p = np.asarray(points, dtype=np.float64)
    n = p.shape[0]
    out = np.full((n, k), np.inf)
    for s in range(0, n, 2048):
        d = ((p[s:s + 2048, None, :] - p[None, :, :]) ** 2).sum(-1)
        d[np.arange(d.shape[0]), np.arange(s, s + d.shape[0])] = np.inf      # the point itself, by index
        d.sort(axis=1)
        m = min(k, n - 1)
        out[s:s + 2048, :m] = d[:, :m]
    return out


def area_downscale(img_u8, out_h, out_w, as_float=False):
    """cv2.resize(..., interpolation=INTER_AREA) as defined: every output pixel is the mean of the
    source area [x W/w, (x+1) W/w) x [y H/h, (y+1) H/h) with fractional coverage weights
    (reference main.py:472-475), the weights of OpenCV's computeResizeAreaTab.  cv2 is absent from
    this image; pinned by PIL's BOX filter for integer factors (where both are the block mean; PIL
    samples by pixel centres otherwise) and by the closed form for a ramp at fractional factors."""
    a = np.asarray(img_u8, dtype=np.float64)
    H, W = a.shape[:2]

    def weights(n_in, n_out):
        m = np.zeros((n_out, n_in))
        s = n_in / n_out
        for o in range(n_out):
            f0, f1 = o * s, min((o + 1) * s, n_in)
            for i in range(int(np.floor(f0)), min(int(np.ceil(f1)), n_in)):
                m[o, i] = min(i + 1, f1) - max(i, f0)
            m[o] /= (f1 - f0)
        return m

    wy, wx = weights(H, out_h), weights(W, out_w)
    r = np.tensordot(wy, a, axes=(1, 0))                       # [y, X, c]  (rows first, then columns:
    r = np.tensordot(r, wx, axes=(1, 1)).transpose(0, 2, 1)    # [y, x, c]   two small products, not one 5-index sum)
    if as_float:
        return r / 255.0
    return np.clip(np.rint(r), 0, 255).astype(np.uint8)


def undistort_image(img_u8, k_src, k_dst, dist, out_h, out_w, fisheye=False):
    """cv2.undistort semantics in float64: output (u, v) <- bilinear sample of the source at
    K_src * distort(K_dst^-1 (u, v)), zero outside (OpenCV initUndistortRectifyMap + remap
    INTER_LINEAR / BORDER_CONSTANT, without remap's 1/32-pixel coefficient quantisation).  Returns
    float64 in [0, 255]."""
    a = np.asarray(img_u8, dtype=np.float64)
    H, W = a.shape[:2]
    d = list(dist) + [0.0] * (8 - len(dist))
    u, v = np.meshgrid(np.arange(out_w, dtype=np.float64), np.arange(out_h, dtype=np.float64))
    x, y = (u - k_dst[2]) / k_dst[0], (v - k_dst[3]) / k_dst[1]
    if not fisheye:
        r2 = x * x + y * y
        rad = (1 + r2 * (d[0] + r2 * (d[1] + r2 * d[4]))) / (1 + r2 * (d[5] + r2 * (d[6] + r2 * d[7])))
        xd = x * rad + 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x)
        yd = y * rad + d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
    else:
        r = np.sqrt(x * x + y * y)
        th = np.arctan(r)
        t2 = th * th
        thd = th * (1 + t2 * (d[0] + t2 * (d[1] + t2 * (d[2] + t2 * d[3]))))
        sc = np.where(r > 1e-8, thd / np.maximum(r, 1e-300), 1.0)
        xd, yd = x * sc, y * sc
    us, vs = k_src[0] * xd + k_src[2], k_src[1] * yd + k_src[3]
    x0, y0 = np.floor(us).astype(np.int64), np.floor(vs).astype(np.int64)
    ax, ay = (us - x0)[..., None], (vs - y0)[..., None]
    out = np.zeros((out_h, out_w, a.shape[2]))
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            wgt = (ax if dx else 1 - ax) * (ay if dy else 1 - ay)
            out += np.where(ok[..., None], wgt * a[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)
    return out
