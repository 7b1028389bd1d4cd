"""Dataset side of `Train-Stage1`: COLMAP model + images -> cameras, targets, initial Gaussians.

What nerfstudio's ColmapDataParser / FullImageDatamanager and SplatfactoModel's initialiser do
upstream of the hot path, reached by the reference through
`ns-train ... colmap --data D --downscale-factor k` (source/container/src/main.py:1303-1306):

  * layout: D/colmap/sparse/0/*.bin (or D/sparse/0), D/images or D/images_{k} (the reference
    pre-creates the downscaled directory, main.py:419-481);
  * world frame = the one the upstream trainer works and EXPORTS in, because the reference's later
    stages assume it (rotate_splat x:270,y:180,z:0 main.py:1481-1500, mirror x :1510-1523, ...):
      frame="nerfstudio" (ns-train ... colmap): ColmapDataParser's axis swap (x, z, -y) =
        `applied_transform`, orientation_method "up" (mean camera up -> +z), center_method "poses"
        (mean camera position -> origin), auto_scale_poses (largest |camera coordinate| -> 1);
      frame="gsplat" (simple_trainer.py): the example Parser's normalisation = similarity_from_cameras
        (mean camera up -> -y, focus point -> origin, median camera distance -> 1) followed by
        align_principle_axes (PCA of the points);
      frame="colmap": centre + scale only (round 1; kept for tests that want the raw axes).
    [UPSTREAM-UNVERIFIED restatements, SURVEY.md 8a row a8]
  * eval split: every 8th image (nerfstudio / gsplat default);
  * Gaussians: means = SfM points, log-scales = log(mean distance to the 3 nearest
    neighbours), random unit quaternions, opacity logit(0.1), SH DC = (rgb - 0.5) / C0,
    higher bands 0  [UPSTREAM-UNVERIFIED defaults, SURVEY.md Appendix A].
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import io_colmap
from . import undistort as ud

SH_C0 = 0.28209479177387814
IMAGE_EXTS = (".jpg", ".jpeg", ".png", ".JPG", ".JPEG", ".PNG")


@dataclass
class Dataset:
    viewmats: torch.Tensor            # [V,4,4] world(normalised)->camera, float32
    Ks: torch.Tensor                  # [V,3,3]
    image_paths: List[str]
    names: List[str]
    width: int
    height: int
    points: torch.Tensor              # [P,3] normalised frame
    points_rgb: torch.Tensor          # [P,3] uint8
    center: torch.Tensor              # frame "colmap": x_norm = (x_world - center) * scale
    scale: float
    train_idx: List[int] = field(default_factory=list)
    eval_idx: List[int] = field(default_factory=list)
    undistort: List[Optional[object]] = field(default_factory=list)   # per image: undistort.Plan or None
    frame: str = "colmap"
    world_rot: Optional[torch.Tensor] = None      # x_norm = scale * world_rot @ x_world + world_shift   (all frames)
    world_shift: Optional[torch.Tensor] = None

    def load_images(self, idx: List[int], device, as_u8: bool = False) -> torch.Tensor:
        """[len(idx),H,W,3] on `device`: float32 in [0,1], or uint8 (`as_u8`, the device image
        cache: 500 4K frames are 12 GB of the 288).  Files are decoded on the host and cross PCIe
        once as uint8; a file larger than (width, height) is area-averaged down ON the GPU
        (ops.image_downscale_area = the INTER_AREA pre-pass of main.py:419-481, without the
        `images_{k}/` directory)."""
        from PIL import Image
        from . import ops
        out = torch.empty(len(idx), self.height, self.width, 3, dtype=torch.uint8 if as_u8 else torch.float32,
                          device=device)
        for j, i in enumerate(idx):
            im = Image.open(self.image_paths[i]).convert("RGB")
            a = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).to(device)
            plan = self.undistort[i] if self.undistort else None
            if plan is not None:
                # (area-average to the size the intrinsics were scaled to, then) resample through the lens model
                if im.size != plan.src_size:
                    if im.size[0] < plan.src_size[0] or im.size[1] < plan.src_size[1]:
                        raise ValueError(f"{self.image_paths[i]}: {im.size} is smaller than the camera's {plan.src_size}")
                    a = ops.image_downscale_area(a, plan.src_size[1], plan.src_size[0])
                out[j] = ops.image_undistort(a, plan.k_src, plan.k_dst, plan.dist, self.height, self.width,
                                             fisheye=plan.fisheye, as_float=not as_u8)
                continue
            if im.size != (self.width, self.height):
                if im.size[0] < self.width or im.size[1] < self.height:
                    raise ValueError(f"{self.image_paths[i]}: {im.size} is smaller than the camera's "
                                     f"{(self.width, self.height)}")
                out[j] = ops.image_downscale_area(a, self.height, self.width, as_float=not as_u8)
            else:
                out[j] = a if as_u8 else a.float().div_(255.0)
        return out

    def denormalise(self, means: torch.Tensor, log_scales: torch.Tensor):
        """Normalised frame -> the COLMAP world frame of the input.  Positions and sizes only: valid as a whole
        for frame="colmap" (no rotation); the upstream exporters do NOT do this, they write the training frame."""
        if self.world_rot is not None and not torch.equal(self.world_rot, torch.eye(3)):
            raise ValueError("denormalise() is for the un-rotated 'colmap' frame; rotate with mi3dgs.transform instead")
        c = self.center.to(means.device, means.dtype)
        return means / self.scale + c, log_scales - math.log(self.scale)

    @property
    def scene_scale(self) -> float:
        """gsplat Parser.scene_scale: largest distance of a camera from the mean camera position."""
        c2w = torch.linalg.inv(self.viewmats.double())
        pos = c2w[:, :3, 3]
        return float((pos - pos.mean(0)).norm(dim=1).max())


APPLIED_TRANSFORM = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, -1.0, 0.0]])   # ColmapDataParser: (x, z, -y)


def rotation_matrix_between(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """nerfstudio camera_utils.rotation_matrix_between: the rotation about a x b that takes a to b (Rodrigues)."""
    a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
    v = np.cross(a, b)
    eps = 1e-6
    if np.abs(v).sum() < eps:
        x = np.array([1.0, 0.0, 0.0]) if abs(a[0]) < eps else np.array([0.0, 1.0, 0.0])
        v = np.cross(a, x)
    v = v / np.linalg.norm(v)
    K = np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    theta = math.acos(float(np.clip(np.dot(a, b), -1.0, 1.0)))
    return np.eye(3) + math.sin(theta) * K + (1.0 - math.cos(theta)) * (K @ K)


def nerfstudio_frame(c2w_cv: np.ndarray):
    """(s, Q, t) with x_n = s Q x_w + t: ColmapDataParser + auto_orient_and_center_poses("up", "poses") +
    auto_scale_poses, from OpenCV camera-to-world matrices [V,4,4].  [UPSTREAM-UNVERIFIED]"""
    P = APPLIED_TRANSFORM
    up_gl = -c2w_cv[:, :3, 1]                          # OpenGL camera +y (up) = minus OpenCV +y (down)
    up = (P @ up_gl.T).T.mean(0)
    up = up / np.linalg.norm(up)
    Rot = rotation_matrix_between(up, np.array([0.0, 0.0, 1.0]))
    origins = (P @ c2w_cv[:, :3, 3].T).T
    mean_o = origins.mean(0)
    pos = (Rot @ (origins - mean_o).T).T
    s = 1.0 / max(float(np.abs(pos).max()), 1e-9)
    Q = Rot @ P
    return s, Q, -s * (Rot @ mean_o)


def gsplat_frame(c2w_cv: np.ndarray, points: np.ndarray):
    """(s, Q, t) of gsplat's examples/datasets Parser(normalize=True): similarity_from_cameras (mean camera up
    -> -y, median closest point of the optical axes to the origin -> origin, median camera distance -> 1), then
    align_principle_axes on the transformed points (PCA, largest variance on x).  [UPSTREAM-UNVERIFIED]"""
    t, R = c2w_cv[:, :3, 3], c2w_cv[:, :3, :3]
    ups = (R * np.array([0.0, -1.0, 0.0])).sum(-1)
    world_up = ups.mean(0)
    world_up = world_up / np.linalg.norm(world_up)
    up_cam = np.array([0.0, -1.0, 0.0])
    c = float((up_cam * world_up).sum())
    cr = np.cross(world_up, up_cam)
    skew = np.array([[0.0, -cr[2], cr[1]], [cr[2], 0.0, -cr[0]], [-cr[1], cr[0], 0.0]])
    R_align = np.eye(3) + skew + (skew @ skew) / (1.0 + c) if c > -1 else np.diag([-1.0, 1.0, 1.0])
    Rr = R_align @ R
    fwds = (Rr * np.array([0.0, 0.0, 1.0])).sum(-1)
    tt = (R_align @ t.T).T
    nearest = tt + (fwds * -tt).sum(-1)[:, None] * fwds
    translate = -np.median(nearest, axis=0)
    s = 1.0 / max(float(np.median(np.linalg.norm(tt + translate, axis=-1))), 1e-9)
    Q1, t1 = R_align, s * translate                      # x1 = s (R_align x + translate)
    p1 = s * (Q1 @ points.T).T + t1
    centroid = np.median(p1, axis=0)
    w, v = np.linalg.eigh(np.cov(p1 - centroid, rowvar=False))
    v = v[:, w.argsort()[::-1]]
    if np.linalg.det(v) < 0:
        v[:, 0] *= -1
    R2 = v.T
    return s, R2 @ Q1, R2 @ t1 - R2 @ centroid


def load_colmap_dataset(data_dir: str, downscale_factor: int = 1, test_every: int = 8,
                        normalize: bool = True, undistort: bool = True, frame: str = "colmap") -> Dataset:
    sparse = io_colmap.find_sparse_dir(data_dir)
    cams = io_colmap.read_cameras(os.path.join(sparse, "cameras.bin"))
    imgs = io_colmap.read_images(os.path.join(sparse, "images.bin"))
    xyz, rgb, _ = io_colmap.read_points3d(os.path.join(sparse, "points3D.bin"))
    k = max(1, int(downscale_factor))
    img_dir = os.path.join(data_dir, "images" if k == 1 else f"images_{k}")
    if not os.path.isdir(img_dir):
        img_dir = os.path.join(data_dir, "images")     # fall back to full size and scale on load
    items = sorted(imgs.values(), key=lambda im: im.name)
    items = [im for im in items if os.path.isfile(os.path.join(img_dir, im.name))]
    if not items:
        raise FileNotFoundError(f"no image listed in {sparse}/images.bin exists under {img_dir}")
    w2c = np.stack([im.world_to_camera() for im in items])                  # [V,4,4]
    c2w = np.linalg.inv(w2c)
    centres = c2w[:, :3, 3]
    if frame not in ("colmap", "nerfstudio", "gsplat"):
        raise ValueError(f"unknown world frame {frame!r}")
    center = centres.mean(0) if normalize else np.zeros(3)
    scale, Q = 1.0, np.eye(3)
    if normalize and frame == "colmap":
        scale = 1.0 / max(float(np.abs(centres - center).max()), 1e-9)
        shift = -scale * center
    elif normalize and frame == "nerfstudio":
        scale, Q, shift = nerfstudio_frame(c2w)
    elif normalize:
        scale, Q, shift = gsplat_frame(c2w, np.asarray(xyz, dtype=np.float64))
    else:
        shift = np.zeros(3)
    # x_n = s Q x_w + shift  =>  x_c = R x_w + t = R Q^T (x_n - shift) / s + t.  Camera space is scaled by s as
    # well, so depths shrink with the scene:  s x_c = (R Q^T) x_n + (s t - R Q^T shift)
    R = w2c[:, :3, :3]
    Rn = np.einsum("vij,kj->vik", R, Q)
    t_n = scale * w2c[:, :3, 3] - np.einsum("vij,j->vi", Rn, shift)
    vm = np.tile(np.eye(4), (len(items), 1, 1))
    vm[:, :3, :3] = Rn
    vm[:, :3, 3] = t_n
    Ks, dist_warn, plans, plan_of = [], False, [], {}
    cam0 = cams[items[0].camera_id]
    W, H = max(1, int(cam0.width / k)), max(1, int(cam0.height / k))       # reference: max(1, int(w / k)), main.py:452
    for im in items:
        c = cams[im.camera_id]
        fx, fy, cx, cy = c.pinhole()
        sx, sy = W / c.width, H / c.height
        Kc = (fx * sx, fy * sy, cx * sx, cy * sy)
        if undistort and im.camera_id not in plan_of:
            plan_of[im.camera_id] = ud.make_plan(c.model, c.distortion(), Kc, (W, H))
        plan = plan_of.get(im.camera_id) if undistort else None
        if plan is not None:
            Kc = plan.K_out
        elif np.abs(c.distortion()).max(initial=0.0) > 1e-6:
            dist_warn = True
        plans.append(plan)
        Ks.append([[Kc[0], 0, Kc[2]], [0, Kc[1], Kc[3]], [0, 0, 1]])
    if dist_warn:
        print("[mi3dgs] note: lens distortion parameters present; images are treated as pinhole (undistort=False)")
    out_sizes = {p.out_size if p is not None else (W, H) for p in plans}
    if len(out_sizes) != 1:
        raise ValueError(f"cameras undistort to different image sizes {sorted(out_sizes)}; one size per dataset is supported")
    src_W, src_H = W, H
    W, H = out_sizes.pop()
    if any(p is not None for p in plans):
        print(f"[mi3dgs] undistorting on the GPU: {src_W}x{src_H} -> {W}x{H} pinhole images")
    V = len(items)
    eval_idx = [i for i in range(V) if test_every > 0 and i % test_every == 0]
    train_idx = [i for i in range(V) if i not in set(eval_idx)] or list(range(V))
    pts = scale * (np.asarray(xyz, dtype=np.float64) @ Q.T) + shift
    return Dataset(torch.from_numpy(vm).float(), torch.tensor(Ks, dtype=torch.float32),
                   [os.path.join(img_dir, im.name) for im in items], [im.name for im in items], W, H,
                   torch.from_numpy(pts).float(), torch.from_numpy(rgb.copy()), torch.from_numpy(center).float(),
                   float(scale), train_idx, eval_idx, plans, frame if normalize else "colmap",
                   torch.from_numpy(Q).float(), torch.from_numpy(np.asarray(shift)).float())


def knn_mean_sq_dist(points: torch.Tensor, k: int = 3) -> torch.Tensor:
    """Mean squared distance to the k nearest neighbours (excluding the point itself): the HIP
    grid search `mi3dgs_knn` (csrc/spatial.hip).  GPU tensors only."""
    from . import ops
    d2 = ops.knn(points.float().contiguous(), k)
    ok = torch.isfinite(d2)                               # fewer than k other points: mean of what exists
    cnt = ok.sum(1).clamp(min=1)
    return torch.where(ok, d2, torch.zeros_like(d2)).sum(1) / cnt


def init_gaussians(points: torch.Tensor, rgb_u8: torch.Tensor, init_opacity: float = 0.1, init_scale: float = 1.0,
                   seed: int = 42, knn_d2: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Initial parameters in the checkpoint schema (post_processing/gsplat_pt_to_ply.py:45-73).
    `knn_d2` (mean squared 3-NN distance per point) is computed on the GPU unless handed in."""
    dev = points.device
    P = points.shape[0]
    d2 = knn_mean_sq_dist(points, 3) if knn_d2 is None else knn_d2.to(dev)
    scales = torch.log(torch.sqrt(d2).clamp_min(1e-7) * init_scale)[:, None].repeat(1, 3)
    g = torch.Generator().manual_seed(seed)
    quats = torch.rand(P, 4, generator=g).to(dev)
    opac = torch.full((P,), math.log(init_opacity / (1.0 - init_opacity)), device=dev)
    sh0 = ((rgb_u8.to(dev).float() / 255.0 - 0.5) / SH_C0)[:, None, :]
    shN = torch.zeros(P, 15, 3, device=dev)
    return dict(means=points.float().contiguous(), quats=quats.float().contiguous(), scales=scales.float().contiguous(),
                opacities=opac.float(), sh0=sh0.float().contiguous(), shN=shN)


def ensure_downscaled_images(images_dir: str, downscale_factor, device=None) -> int:
    """`ensure_downscaled_images` of the reference (source/container/src/main.py:419-481): creates
    `images_{k}/` beside `images_dir` with every .jpg/.jpeg/.png area-averaged to
    (max(1, int(w/k)), max(1, int(h/k))), skipping the work when the directory is already
    complete.  The resize runs on the GPU; returns the number of files written."""
    from PIL import Image
    from . import ops
    try:
        k = int(str(downscale_factor))
    except (TypeError, ValueError):
        k = 1
    if k <= 1 or not os.path.isdir(images_dir):
        return 0
    target = os.path.join(os.path.dirname(images_dir), f"images_{k}")
    os.makedirs(target, exist_ok=True)
    exts = (".jpg", ".jpeg", ".png")
    files = sorted(f for f in os.listdir(images_dir) if f.lower().endswith(exts))
    if not files or len([f for f in os.listdir(target) if f.lower().endswith(exts)]) == len(files):
        return 0
    dev = torch.device(device if device is not None else "cuda")
    n = 0
    for f in files:
        im = Image.open(os.path.join(images_dir, f))
        if im.mode not in ("L", "RGB", "RGBA"):
            im = im.convert("RGB")
        a = np.asarray(im, dtype=np.uint8)
        a = a[:, :, None] if a.ndim == 2 else a
        h, w = a.shape[:2]
        out = ops.image_downscale_area(torch.from_numpy(a.copy()).to(dev), max(1, int(h / k)), max(1, int(w / k)))
        o = out.cpu().numpy()
        Image.fromarray(o[:, :, 0] if o.shape[2] == 1 else o).save(os.path.join(target, f))
        n += 1
    return n
