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

    def load_images(self, idx: List[int], device) -> torch.Tensor:
        """[len(idx),H,W,3] float32 in [0,1] on `device` (decoded on the host, one PCIe trip)."""
        from PIL import Image
        out = torch.empty(len(idx), self.height, self.width, 3, dtype=torch.float32, device=device)
        for j, i in enumerate(idx):
            im = Image.open(self.image_paths[i]).convert("RGB")
            if im.size != (self.width, self.height):
                im = im.resize((self.width, self.height), Image.BOX)
            a = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy())
            out[j] = a.to(device).float().div_(255.0)
        return out

    def denormalise(self, means: torch.Tensor, log_scales: torch.Tensor):
        """Normalised frame -> the COLMAP world frame of the input (for export)."""
        c = self.center.to(means.device, means.dtype)
        return means / self.scale + c, log_scales - math.log(self.scale)


def load_colmap_dataset(data_dir: str, downscale_factor: int = 1, test_every: int = 8,
                        normalize: bool = True) -> Dataset:
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
    Ks, dist_warn = [], False
    cam0 = cams[items[0].camera_id]
    W, H = max(1, int(cam0.width / k)), max(1, int(cam0.height / k))       # reference: max(1, int(w / k)), main.py:452
    for im in items:
        c = cams[im.camera_id]
        fx, fy, cx, cy = c.pinhole()
        sx, sy = W / c.width, H / c.height
        Ks.append([[fx * sx, 0, cx * sx], [0, fy * sy, cy * sy], [0, 0, 1]])
        if np.abs(c.distortion()).max(initial=0.0) > 1e-6:
            dist_warn = True
    if dist_warn:
        print("[mi3dgs] note: lens distortion parameters present; images are treated as pinhole "
              "(the reference's multi-GPU branch undistorts first, main.py:1157-1180)")
    V = len(items)
    eval_idx = [i for i in range(V) if test_every > 0 and i % test_every == 0]
    train_idx = [i for i in range(V) if i not in set(eval_idx)] or list(range(V))
    pts = (xyz - center) * scale
    return Dataset(torch.from_numpy(vm).float(), torch.tensor(Ks, dtype=torch.float32),
                   [os.path.join(img_dir, im.name) for im in items], [im.name for im in items], W, H,
                   torch.from_numpy(pts).float(), torch.from_numpy(rgb.copy()), torch.from_numpy(center).float(),
                   float(scale), train_idx, eval_idx)


def knn_mean_sq_dist(points: torch.Tensor, k: int = 3, chunk: int = 4096) -> torch.Tensor:
    """Mean squared distance to the k nearest neighbours (excluding the point itself).
    Chunked brute force on whatever device `points` lives on."""
    P = points.shape[0]
    out = torch.empty(P, dtype=points.dtype, device=points.device)
    kk = min(k + 1, P)
    for s in range(0, P, chunk):
        d2 = torch.cdist(points[s:s + chunk], points).pow_(2)
        v = torch.topk(d2, kk, dim=1, largest=False).values[:, 1:]
        out[s:s + chunk] = v.mean(1) if v.numel() else 0.0
    return out


def init_gaussians(points: torch.Tensor, rgb_u8: torch.Tensor, init_opacity: float = 0.1, init_scale: float = 1.0,
                   seed: int = 42) -> Dict[str, torch.Tensor]:
    """Initial parameters in the checkpoint schema (post_processing/gsplat_pt_to_ply.py:45-73)."""
    dev = points.device
    P = points.shape[0]
    d2 = knn_mean_sq_dist(points, 3)
    scales = torch.log(torch.sqrt(d2).clamp_min(1e-7) * init_scale)[:, None].repeat(1, 3)
    g = torch.Generator().manual_seed(seed)
    quats = torch.rand(P, 4, generator=g).to(dev)
    opac = torch.full((P,), math.log(init_opacity / (1.0 - init_opacity)), device=dev)
    sh0 = ((rgb_u8.to(dev).float() / 255.0 - 0.5) / SH_C0)[:, None, :]
    shN = torch.zeros(P, 15, 3, device=dev)
    return dict(means=points.float().contiguous(), quats=quats.float().contiguous(), scales=scales.float().contiguous(),
                opacities=opac.float(), sh0=sh0.float().contiguous(), shN=shN)
