"""Dataset side of `Train-Stage1`: COLMAP model + images -> cameras, targets, initial Gaussians.

What nerfstudio's ColmapDataParser / FullImageDatamanager and SplatfactoModel's initialiser do
upstream of the hot path, reached by the reference through
`ns-train ... colmap --data D --downscale-factor k` (source/container/src/main.py:1303-1306):

  * layout: D/colmap/sparse/0/*.bin (or D/sparse/0), D/images or D/images_{k} (the reference
    pre-creates the downscaled directory, main.py:419-481);
  * poses: translated to the centroid of the camera centres and scaled so the farthest camera
    sits at distance 1 (nerfstudio "center_method=poses" + auto_scale_poses; no re-orientation,
    the reference re-orients the exported PLY itself, main.py:1481-1500);
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
    center: torch.Tensor              # normalisation: x_norm = (x_world - center) * scale
    scale: float
    train_idx: List[int] = field(default_factory=list)
    eval_idx: List[int] = field(default_factory=list)
    undistort: List[Optional[object]] = field(default_factory=list)   # per image: undistort.Plan or None

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
        """Normalised frame -> the COLMAP world frame of the input (for export)."""
        c = self.center.to(means.device, means.dtype)
        return means / self.scale + c, log_scales - math.log(self.scale)


def load_colmap_dataset(data_dir: str, downscale_factor: int = 1, test_every: int = 8,
                        normalize: bool = True, undistort: bool = True) -> Dataset:
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
    center = centres.mean(0) if normalize else np.zeros(3)
    scale = 1.0
    if normalize:
        scale = 1.0 / max(float(np.abs(centres - center).max()), 1e-9)
    # x_n = (x_w - center) * scale  =>  camera = R x_w + t = R (x_n / scale + center) + t ; keep metric
    # camera space scaled by `scale` as well so depths shrink with the scene: t_n = (R center + t) * scale
    R = w2c[:, :3, :3]
    t_n = (np.einsum("vij,j->vi", R, center) + w2c[:, :3, 3]) * scale
    vm = np.tile(np.eye(4), (len(items), 1, 1))
    vm[:, :3, :3] = R
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
    pts = (xyz - center) * scale
    return Dataset(torch.from_numpy(vm).float(), torch.tensor(Ks, dtype=torch.float32),
                   [os.path.join(img_dir, im.name) for im in items], [im.name for im in items], W, H,
                   torch.from_numpy(pts).float(), torch.from_numpy(rgb.copy()), torch.from_numpy(center).float(),
                   float(scale), train_idx, eval_idx, plans)


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
