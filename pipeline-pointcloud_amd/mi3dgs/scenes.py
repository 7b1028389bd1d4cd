"""Seeded synthetic scenes S0-S3 (SURVEY.md 8d / BASELINE.md section 2).

There is no network for datasets, so benchmarks and tests run on these generators:
  S0 "cube"        N=10 000   4 cameras  800x800   fx 960   (BASELINE.json configs[0])
  S1 "lego-like"   N=300 000  100 cameras 800x800  fx 1111  (configs[1])
  S2 "garden-like" N=2 000 000 185 cameras 1920x1080 fx 1450 SH deg 3 (configs[2], the metric's config)
  S3 "6M"          N=6 000 000 same generator as S2, radius 20       (configs[4])
Everything is generated on the CPU with torch.Generator(seed) so it is bit-reproducible,
then moved to the requested device.  Parameters are returned in *parameter space*
(log scales, logit opacities), the layout of the reference checkpoint schema
(post_processing/gsplat_pt_to_ply.py:45-73).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch


@dataclass
class Scene:
    name: str
    params: Dict[str, torch.Tensor]     # means[N,3] quats[N,4] scales[N,3](log) opacities[N](logit) sh0[N,1,3] shN[N,15,3]
    viewmats: torch.Tensor              # [V,4,4] world->camera (x right, y down, z forward)
    Ks: torch.Tensor                    # [V,3,3]
    width: int
    height: int
    sh_degree: int = 3

    def to(self, device) -> "Scene":
        return Scene(self.name, {k: v.to(device) for k, v in self.params.items()}, self.viewmats.to(device),
                     self.Ks.to(device), self.width, self.height, self.sh_degree)


def look_at(eye: torch.Tensor, target: torch.Tensor, up=(0.0, 0.0, 1.0)) -> torch.Tensor:
    """World->camera matrix of a camera at `eye` looking at `target` (OpenCV axes)."""
    up = torch.tensor(up, dtype=torch.float64)
    f = (target - eye).to(torch.float64)
    f = f / f.norm()
    r = torch.linalg.cross(f, up)
    r = r / r.norm()
    d = torch.linalg.cross(f, r)
    R = torch.stack([r, d, f])
    V = torch.eye(4, dtype=torch.float64)
    V[:3, :3] = R
    V[:3, 3] = -R @ eye.to(torch.float64)
    return V.to(torch.float32)


def _intrinsics(fx: float, w: int, h: int) -> torch.Tensor:
    return torch.tensor([[fx, 0.0, w / 2.0], [0.0, fx, h / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float32)


def _params(means: torch.Tensor, log_scales: torch.Tensor, g: torch.Generator) -> Dict[str, torch.Tensor]:
    N = means.shape[0]
    quats = torch.randn(N, 4, generator=g)
    quats = quats / quats.norm(dim=-1, keepdim=True)
    opac = torch.rand(N, generator=g) * 4.0 - 2.0
    sh0 = (torch.rand(N, 1, 3, generator=g) * 2.0 - 1.0)
    shN = torch.randn(N, 15, 3, generator=g) * 0.1
    return dict(means=means.float().contiguous(), quats=quats.float().contiguous(),
                scales=log_scales.float().contiguous(), opacities=opac.float().contiguous(),
                sh0=sh0.float().contiguous(), shN=shN.float().contiguous())


def ring_cameras(n: int, radius: float, h_lo: float, h_hi: float, fx: float, w: int, h: int, g: torch.Generator):
    vms, ks = [], []
    for i in range(n):
        a = 2.0 * math.pi * i / n
        z = h_lo + (h_hi - h_lo) * float(torch.rand(1, generator=g))
        eye = torch.tensor([radius * math.cos(a), radius * math.sin(a), z])
        vms.append(look_at(eye, torch.zeros(3)))
        ks.append(_intrinsics(fx, w, h))
    return torch.stack(vms), torch.stack(ks)


def make_cube(n: int = 10_000, seed: int = 0, width: int = 800, height: int = 800, n_views: int = 4,
              fx: float = 960.0) -> Scene:
    g = torch.Generator().manual_seed(seed)
    means = torch.rand(n, 3, generator=g) * 2.0 - 1.0
    ls = torch.rand(n, 3, generator=g) * (math.log(0.05) - math.log(0.01)) + math.log(0.01)
    P = _params(means, ls, g)
    elev = math.radians(20.0)
    vms, ks = [], []
    for i in range(n_views):
        a = 2.0 * math.pi * i / n_views
        eye = torch.tensor([4.0 * math.cos(a) * math.cos(elev), 4.0 * math.sin(a) * math.cos(elev), 4.0 * math.sin(elev)])
        vms.append(look_at(eye, torch.zeros(3)))
        ks.append(_intrinsics(fx, width, height))
    return Scene("S0-cube", P, torch.stack(vms), torch.stack(ks), width, height)


def make_lego_like(n: int = 300_000, seed: int = 1, width: int = 800, height: int = 800, n_views: int = 100,
                   fx: float = 1111.0) -> Scene:
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    r = 0.6 + 0.4 * torch.rand(n, 1, generator=g)
    means = d * r
    ls = torch.rand(n, 3, generator=g) * (math.log(0.02) - math.log(0.004)) + math.log(0.004)
    P = _params(means, ls, g)
    vms, ks = [], []
    for i in range(n_views):
        u = (i + 0.5) / n_views
        phi = math.acos(1.0 - u * 0.95)            # upper hemisphere
        th = math.pi * (1 + 5 ** 0.5) * i
        eye = torch.tensor([4.0 * math.sin(phi) * math.cos(th), 4.0 * math.sin(phi) * math.sin(th), 4.0 * math.cos(phi)])
        up = (0.0, 0.0, 1.0) if phi > 0.15 else (0.0, 1.0, 0.0)
        vms.append(look_at(eye, torch.zeros(3), up))
        ks.append(_intrinsics(fx, width, height))
    return Scene("S1-lego-like", P, torch.stack(vms), torch.stack(ks), width, height)


def make_garden_like(n: int = 2_000_000, seed: int = 2, width: int = 1920, height: int = 1080, n_views: int = 185,
                     fx: float = 1450.0, ground_radius: float = 8.0, cam_radius: float = 5.0,
                     name: Optional[str] = None) -> Scene:
    g = torch.Generator().manual_seed(seed)
    n_ground = int(0.6 * n)
    n_blob = n - n_ground
    rr = ground_radius * torch.sqrt(torch.rand(n_ground, generator=g))
    aa = 2.0 * math.pi * torch.rand(n_ground, generator=g)
    ground = torch.stack([rr * torch.cos(aa), rr * torch.sin(aa), 0.05 * torch.randn(n_ground, generator=g)], dim=-1)
    d = torch.randn(n_blob, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    blob = d * (1.5 * torch.rand(n_blob, 1, generator=g) ** (1.0 / 3.0))
    blob[:, 2] += 0.8
    means = torch.cat([ground, blob], dim=0)
    perm = torch.randperm(n, generator=g)
    means = means[perm]
    ls = math.log(0.02) + 0.7 * torch.randn(n, 3, generator=g)
    P = _params(means, ls, g)
    vms, ks = ring_cameras(n_views, cam_radius, 1.5, 2.5, fx, width, height, g)
    return Scene(name or "S2-garden-like", P, vms, ks, width, height)


def make_6m(n: int = 6_000_000, seed: int = 3, **kw) -> Scene:
    return make_garden_like(n=n, seed=seed, ground_radius=20.0, cam_radius=5.0, name="S3-6M", **kw)


def add_backdrop(sc: Scene, n: int, radius: float, centre=(0.0, 0.0, 0.0), seed: int = 11) -> Scene:
    """The scene inside an opaque shell of `n` Gaussians of radius `radius` around `centre`, smoothly coloured by direction:
    every pixel of every view then shows scene content, as in a photograph.  (The end-to-end training datasets need it: RGB
    targets whose uncovered pixels carry a constant background colour can only be fitted by a fog of huge transparent
    Gaussians -- not a thing a trainer should be validated on; tools/train_synthetic.py, tools/train_wolf.py.)"""
    g = torch.Generator().manual_seed(seed)
    i = torch.arange(n, dtype=torch.float64) + 0.5
    phi = torch.acos(1.0 - 2.0 * i / n)                       # Fibonacci sphere: even spacing
    th = math.pi * (1.0 + 5.0 ** 0.5) * i
    d = torch.stack([torch.sin(phi) * torch.cos(th), torch.sin(phi) * torch.sin(th), torch.cos(phi)], dim=-1).float()
    means = torch.tensor(centre, dtype=torch.float32) + radius * d
    spacing = radius * math.sqrt(4.0 * math.pi / n)
    ls = torch.full((n, 3), math.log(0.8 * spacing))
    quats = torch.zeros(n, 4)
    quats[:, 0] = 1.0
    # low-frequency colour pattern in [0.15, 0.85], as SH DC:  rgb = 0.5 + 0.2821 * sh0
    rgb = 0.5 + 0.35 * torch.stack([torch.sin(3.0 * d[:, 0] + 1.0) * torch.cos(2.0 * d[:, 1]),
                                    torch.sin(2.0 * d[:, 1] + 2.0) * torch.cos(3.0 * d[:, 2]),
                                    torch.sin(4.0 * d[:, 2]) * torch.cos(2.0 * d[:, 0] + 0.5)], dim=-1)
    rgb = rgb + 0.03 * torch.randn(n, 3, generator=g)
    B = dict(means=means, quats=quats, scales=ls, opacities=torch.full((n,), 4.0), sh0=((rgb - 0.5) / 0.28209479)[:, None, :],
             shN=torch.zeros(n, 15, 3))
    P = {k: torch.cat([sc.params[k], B[k].to(sc.params[k].dtype)], dim=0).contiguous() for k in sc.params}
    return Scene(sc.name + "+backdrop", P, sc.viewmats, sc.Ks, sc.width, sc.height, sc.sh_degree)


def make_scene(kind: str, **kw) -> Scene:
    return {"cube": make_cube, "lego": make_lego_like, "garden": make_garden_like, "6m": make_6m}[kind](**kw)
