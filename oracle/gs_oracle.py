"""CPU oracle for the 3D Gaussian Splatting hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference repository (krishan44/pipeline-pointcloud) holds no
rasteriser, no tests and no golden vectors for this path: it reaches the arithmetic only
by shelling out to unpinned third-party code --
  source/container/src/main.py:1270-1316   `ns-train splatfacto ... colmap --data D`
  source/container/src/main.py:1318-1347   `gsplat/examples/simple_trainer.py default ...`
  source/container/Dockerfile:215-235      nerfstudio / gsplat cloned at `main`, no tag.
gsplat and nerfstudio are absent from /root/reference and from this image, so this file
restates their *published* algorithm (Kerbl et al. 2023, "3D Gaussian Splatting for
Real-Time Radiance Field Rendering"; Ye et al. 2024, "gsplat", arXiv 2409.06765) and is
anchored on what the reference itself pins: the call sites above, the checkpoint schema
(post_processing/gsplat_pt_to_ply.py:45-73) and the PLY field order
(post_processing/spz/src/cc/load-spz.cc:572-750).

Only `tests/`, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may import this
module.  The product path (pipeline-pointcloud_amd/) never does: it fails loudly when the
HIP library is missing.

Everything is plain PyTorch on CPU, dtype-generic (float64 for gradcheck / golden vectors,
float32 to mirror the kernels), and differentiable through autograd, so the analytic
backward passes of the HIP kernels are checked against an independent derivation.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

ALPHA_THRESHOLD = 1.0 / 255.0      # gsplat: skip a splat whose alpha is below this
MAX_ALPHA = 0.999                  # gsplat: alpha clamp
TRANSMITTANCE_STOP = 1e-4          # gsplat: a pixel is done when T would drop to <= this
EPS2D = 0.3                        # screen-space blur added to the 2-D covariance (px^2)
RADIUS_EXTEND = 3.33               # bounding extent in sigmas (gsplat >= 1.5)

SH_C0 = 0.2820947917738781
SH_C1 = 0.48860251190292
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


# ----------------------------------------------------------------------------- geometry
def quat_to_rotmat(quats: torch.Tensor) -> torch.Tensor:
    """wxyz quaternion (normalised here, as gsplat does inside the kernel) -> [...,3,3]."""
    q = F.normalize(quats, dim=-1)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
    ], dim=-1)
    return R.reshape(quats.shape[:-1] + (3, 3))


def quat_scale_to_covar(quats: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """Sigma = R S S^T R^T, scales already exponentiated.  [N,3,3]."""
    R = quat_to_rotmat(quats)
    M = R * scales[..., None, :]
    return M @ M.transpose(-1, -2)


def projection(means, quats, scales, viewmats, Ks, width: int, height: int,
               opacities: Optional[torch.Tensor] = None, eps2d: float = EPS2D,
               near_plane: float = 0.01, far_plane: float = 1e10,
               radius_clip: float = 0.0, calc_compensations: bool = False):
    """gsplat `fully_fused_projection` (pinhole, EWA), SURVEY.md 2a row 2.

    means[N,3] quats[N,4] scales[N,3] viewmats[C,4,4] (world->camera) Ks[C,3,3]
    -> radii[C,N,2] int32, means2d[C,N,2], depths[C,N], conics[C,N,3], compensations[C,N]|None
    Culled Gaussians have radii == 0; their other outputs are zero.
    """
    C, N = viewmats.shape[0], means.shape[0]
    dt = means.dtype
    R = viewmats[:, :3, :3]
    t = viewmats[:, :3, 3]
    mean_c = torch.einsum("cij,nj->cni", R, means) + t[:, None, :]          # [C,N,3]
    covar = quat_scale_to_covar(quats, scales)                               # [N,3,3]
    covar_c = torch.einsum("cij,njk,clk->cnil", R, covar, R)                 # [C,N,3,3]

    fx, fy = Ks[:, 0, 0][:, None], Ks[:, 1, 1][:, None]
    cx, cy = Ks[:, 0, 2][:, None], Ks[:, 1, 2][:, None]
    x, y, z = mean_c.unbind(-1)
    tan_fovx = 0.5 * width / fx
    tan_fovy = 0.5 * height / fy
    lim_x_pos = (width - cx) / fx + 0.3 * tan_fovx
    lim_x_neg = cx / fx + 0.3 * tan_fovx
    lim_y_pos = (height - cy) / fy + 0.3 * tan_fovy
    lim_y_neg = cy / fy + 0.3 * tan_fovy
    rz = 1.0 / z
    tx = z * torch.minimum(lim_x_pos, torch.maximum(-lim_x_neg, x * rz))
    ty = z * torch.minimum(lim_y_pos, torch.maximum(-lim_y_neg, y * rz))
    zero = torch.zeros_like(z)
    J = torch.stack([fx * rz, zero, -fx * tx * rz * rz,
                     zero, fy * rz, -fy * ty * rz * rz], dim=-1).reshape(C, N, 2, 3)
    cov2d = J @ covar_c @ J.transpose(-1, -2)                                # [C,N,2,2]
    means2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], dim=-1)

    a0, b0, c0 = cov2d[..., 0, 0], cov2d[..., 0, 1], cov2d[..., 1, 1]
    det_orig = a0 * c0 - b0 * b0
    a, c, b = a0 + eps2d, c0 + eps2d, b0
    det = a * c - b * b
    compensation = torch.sqrt(torch.clamp(det_orig / det, min=0.0))
    conics = torch.stack([c / det, -b / det, a / det], dim=-1)

    valid = (z >= near_plane) & (z <= far_plane) & (det > 0)
    with torch.no_grad():
        extend = torch.full_like(z, RADIUS_EXTEND)
        if opacities is not None:
            op = opacities[None, :].expand(C, N).to(dt)
            if calc_compensations:
                op = op * compensation
            valid = valid & (op >= ALPHA_THRESHOLD)
            ext_op = torch.sqrt(2.0 * torch.log(torch.clamp(op, min=1e-30) / ALPHA_THRESHOLD).clamp(min=0))
            extend = torch.minimum(extend, ext_op)
        bmid = 0.5 * (a + c)
        v1 = bmid + torch.sqrt(torch.clamp(bmid * bmid - det, min=0.01))
        r1 = extend * torch.sqrt(v1)
        rx = torch.ceil(torch.minimum(extend * torch.sqrt(a), r1))
        ry = torch.ceil(torch.minimum(extend * torch.sqrt(c), r1))
        valid = valid & ~((rx <= radius_clip) & (ry <= radius_clip))
        mx, my = means2d[..., 0], means2d[..., 1]
        valid = valid & ~((mx + rx <= 0) | (mx - rx >= width) | (my + ry <= 0) | (my - ry >= height))
        radii = torch.stack([rx, ry], dim=-1)
        radii = torch.where(valid[..., None], radii, torch.zeros_like(radii)).to(torch.int32)

    vm = valid[..., None]
    means2d = torch.where(vm, means2d, torch.zeros_like(means2d))
    conics = torch.where(vm, conics, torch.zeros_like(conics))
    depths = torch.where(valid, z, torch.zeros_like(z))
    comp = torch.where(valid, compensation, torch.zeros_like(compensation)) if calc_compensations else None
    return radii, means2d, depths, conics, comp


# ------------------------------------------------------------------- spherical harmonics
def sh_basis(degree: int, dirs: torch.Tensor) -> torch.Tensor:
    """Real SH basis up to `degree` for (normalised here) directions -> [..., (degree+1)^2]."""
    d = F.normalize(dirs, dim=-1)
    x, y, z = d.unbind(-1)
    out = [torch.full_like(x, SH_C0)]
    if degree >= 1:
        out += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if degree >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        out += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy),
                SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if degree >= 3:
        out += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4 * zz - xx - yy),
                SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy),
                SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(out, dim=-1)


def spherical_harmonics(degree: int, dirs: torch.Tensor, coeffs: torch.Tensor) -> torch.Tensor:
    """gsplat `spherical_harmonics`: dirs[...,3], coeffs[...,K,3] -> colours[...,3] (no +0.5)."""
    nb = (degree + 1) ** 2
    basis = sh_basis(degree, dirs)
    return (basis[..., :, None] * coeffs[..., :nb, :]).sum(-2)


# ----------------------------------------------------------------------- tile binning
def isect_tiles(means2d, radii, depths, tile_size: int, tile_width: int, tile_height: int):
    """gsplat `isect_tiles` (+ its radix sort).  Returns
    tiles_per_gauss[C,N] int32, isect_ids[I] int64 (sorted), flatten_ids[I] int32 (sorted).
    Key = ((cam * n_tiles + tile) << 32) | f32 depth bits; ties keep Gaussian-index order.
    """
    C, N = depths.shape
    m = means2d.detach().to(torch.float32)
    r = radii.to(torch.float32)
    ts = float(tile_size)
    tmin_x = torch.clamp(torch.floor((m[..., 0] - r[..., 0]) / ts), 0, tile_width).to(torch.int64)
    tmin_y = torch.clamp(torch.floor((m[..., 1] - r[..., 1]) / ts), 0, tile_height).to(torch.int64)
    tmax_x = torch.clamp(torch.ceil((m[..., 0] + r[..., 0]) / ts), 0, tile_width).to(torch.int64)
    tmax_y = torch.clamp(torch.ceil((m[..., 1] + r[..., 1]) / ts), 0, tile_height).to(torch.int64)
    vis = (radii > 0).all(-1)
    tiles = torch.where(vis, (tmax_x - tmin_x) * (tmax_y - tmin_y), torch.zeros_like(tmin_x))
    n_tiles = tile_width * tile_height
    dbits = depths.detach().to(torch.float32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    # emission order = gsplat's: (camera, Gaussian) major, then tile rows, then tile columns
    flat_tiles = tiles.flatten()
    sel = torch.nonzero(flat_tiles > 0).flatten()
    if sel.numel() > 0:
        cnt = flat_tiles[sel]
        w = (tmax_x - tmin_x).flatten()[sel]
        rep = torch.repeat_interleave(torch.arange(sel.numel()), cnt)
        start = torch.cumsum(cnt, 0) - cnt
        local = torch.arange(int(cnt.sum())) - start[rep]
        ty = tmin_y.flatten()[sel][rep] + local // w[rep]
        tx = tmin_x.flatten()[sel][rep] + local % w[rep]
        cam = (sel // N)[rep]
        tid = cam * n_tiles + ty * tile_width + tx
        keys = (tid << 32) | dbits.flatten()[sel][rep]
        vals = sel[rep]
        order = torch.sort(keys, stable=True).indices
        keys, vals = keys[order], vals[order]
    else:
        keys = torch.zeros(0, dtype=torch.int64)
        vals = torch.zeros(0, dtype=torch.int64)
    return tiles.to(torch.int32), keys, vals.to(torch.int32)


def isect_offset_encode(isect_ids: torch.Tensor, C: int, tile_width: int, tile_height: int):
    """First index of each (camera, tile) in the sorted intersection list.  [C,TH,TW] int32."""
    n_tiles = tile_width * tile_height
    tid = isect_ids >> 32
    bounds = torch.arange(C * n_tiles, dtype=torch.int64)
    return torch.searchsorted(tid.contiguous(), bounds, right=False).to(torch.int32).reshape(
        C, tile_height, tile_width)


# -------------------------------------------------------------------------- compositing
def rasterize_to_pixels(means2d, conics, colors, opacities, width: int, height: int,
                        tile_size: int, isect_offsets, flatten_ids,
                        backgrounds: Optional[torch.Tensor] = None, marginal: Optional[dict] = None):
    """gsplat `rasterize_to_pixels`: front-to-back alpha compositing per tile.

    marginal (checker-side diagnostic, not part of gsplat): a dict that receives "alpha_skip" = bool [C,H,W], true where the
    `alpha < 1/255 -> skip` decision of some splat the pixel reaches is closer to its threshold than float32 can resolve:
    |sigma - ln(255 o)| <= ulps * 2^-24 * (|A| dx^2 / 2 + |C| dy^2 / 2 + |B dx dy|) + 1e-6, ulps = marginal["ulps"] (default 16).
    A float32 implementation (gsplat's kernels, this repository's) may take that decision either way; each flip moves the
    pixel's colour by one splat's alpha >= 1/255, so parity tests weigh such pixels with zero instead of excusing Gaussians.

    means2d[C,N,2] conics[C,N,3] colors[C,N,D] opacities[C,N] -> render[C,H,W,D],
    alphas[C,H,W,1], last_ids[C,H,W] (index into flatten_ids of the last splat composited,
    0 when none).  Per pixel, over the tile's depth-sorted list:
        sigma = .5(A dx^2 + C dy^2) + B dx dy ; alpha = min(.999, o e^-sigma)
        skip if sigma < 0 or alpha < 1/255 ; stop BEFORE a splat that would make T <= 1e-4
    """
    C, N = means2d.shape[:2]
    D = colors.shape[-1]
    dt = means2d.dtype
    TH, TW = isect_offsets.shape[1:]
    n_isect = flatten_ids.shape[0]
    offs = isect_offsets.flatten().tolist() + [n_isect]
    m_f = means2d.reshape(C * N, 2)
    c_f = conics.reshape(C * N, 3)
    col_f = colors.reshape(C * N, D)
    o_f = opacities.reshape(C * N)
    render = torch.zeros(C, height, width, D, dtype=dt)
    alphas = torch.zeros(C, height, width, 1, dtype=dt)
    last_ids = torch.zeros(C, height, width, dtype=torch.int32)
    rows, cols = [], []
    marg_rows = []
    for c in range(C):
        for ty in range(TH):
            for tx in range(TW):
                t = (c * TH + ty) * TW + tx
                s, e = offs[t], offs[t + 1]
                y0, x0 = ty * tile_size, tx * tile_size
                y1, x1 = min(y0 + tile_size, height), min(x0 + tile_size, width)
                if y1 <= y0 or x1 <= x0:
                    continue
                py, px = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
                P = py.numel()
                pxf = px.flatten().to(dt) + 0.5
                pyf = py.flatten().to(dt) + 0.5
                if e > s:
                    ids = flatten_ids[s:e].long()
                    dx = m_f[ids, 0][None, :] - pxf[:, None]
                    dy = m_f[ids, 1][None, :] - pyf[:, None]
                    cn = c_f[ids]
                    sigma = 0.5 * (cn[:, 0] * dx * dx + cn[:, 2] * dy * dy) + cn[:, 1] * dx * dy
                    alpha = torch.clamp(o_f[ids][None, :] * torch.exp(-sigma), max=MAX_ALPHA)
                    keep = (sigma >= 0) & (alpha >= ALPHA_THRESHOLD)
                    alpha = torch.where(keep, alpha, torch.zeros_like(alpha))
                    one_m = 1.0 - alpha
                    T_incl = torch.cumprod(one_m, dim=1)
                    T_excl = torch.cat([torch.ones(P, 1, dtype=dt), T_incl[:, :-1]], dim=1)
                    live = (T_incl > TRANSMITTANCE_STOP).detach()        # prefix-closed
                    if marginal is not None:
                        with torch.no_grad():
                            terms = 0.5 * (cn[:, 0].abs() * dx * dx + cn[:, 2].abs() * dy * dy) + (cn[:, 1] * dx * dy).abs()
                            s_th = torch.log(o_f[ids] / ALPHA_THRESHOLD)[None, :]
                            near = (sigma - s_th).abs() <= float(marginal.get("ulps", 16)) * 2.0 ** -24 * terms + 1e-6
                            reached = torch.cat([torch.ones(P, 1, dtype=torch.bool), live[:, :-1]], dim=1)
                            marg_rows.append((c, py.flatten(), px.flatten(), (near & reached).any(dim=1)))
                    w = alpha * T_excl * live
                    rgb = w @ col_f[ids]
                    T_fin = torch.prod(torch.where(live, one_m, torch.ones_like(one_m)), dim=1)
                    contrib = keep & live
                    idx = torch.arange(s, e)[None, :].expand(P, -1)
                    last = torch.where(contrib, idx, torch.zeros_like(idx)).max(dim=1).values
                else:
                    rgb = torch.zeros(P, D, dtype=dt)
                    T_fin = torch.ones(P, dtype=dt)
                    last = torch.zeros(P, dtype=torch.int64)
                if backgrounds is not None:
                    rgb = rgb + T_fin[:, None] * backgrounds[c][None, :]
                rows.append((c, py.flatten(), px.flatten(), rgb, 1.0 - T_fin, last))
    if rows:
        # one scatter for all tiles (a per-tile in-place write makes autograd carry a full-size image per tile)
        pix = torch.cat([(c * height + py) * width + px for c, py, px, _, _, _ in rows])
        render = render.reshape(-1, D).index_put((pix,), torch.cat([r[3] for r in rows])).reshape(C, height, width, D)
        alphas = alphas.reshape(-1).index_put((pix,), torch.cat([r[4] for r in rows])).reshape(C, height, width, 1)
        last_ids = last_ids.reshape(-1).index_put((pix,), torch.cat([r[5] for r in rows]).to(torch.int32)).reshape(
            C, height, width)
    if marginal is not None:
        m = torch.zeros(C * height * width, dtype=torch.bool)
        if marg_rows:
            m[torch.cat([(c * height + py) * width + px for c, py, px, _ in marg_rows])] = torch.cat([r[3] for r in marg_rows])
        marginal["alpha_skip"] = m.reshape(C, height, width)
    return render, alphas, last_ids


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width: int, height: int,
                  near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
                  eps2d: float = EPS2D, sh_degree: Optional[int] = None, tile_size: int = 16,
                  backgrounds: Optional[torch.Tensor] = None, rasterize_mode: str = "classic",
                  absgrad: bool = False, marginal: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
    """gsplat `rasterization(...)` (SURVEY.md 8b tier 2), unpacked path.

    scales are already exp()'d and opacities already sigmoid()'d, as the callers pass them.
    colors is [N,K,3] SH coefficients when `sh_degree` is given, else [N,D] / [C,N,D] colours.
    """
    C, N = viewmats.shape[0], means.shape[0]
    aa = rasterize_mode == "antialiased"
    radii, means2d, depths, conics, comps = projection(
        means, quats, scales, viewmats, Ks, width, height, opacities=opacities, eps2d=eps2d,
        near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip,
        calc_compensations=aa)
    op = opacities[None, :].expand(C, N)
    if aa:
        op = op * comps
    if sh_degree is not None:
        campos = torch.linalg.inv(viewmats)[:, :3, 3]
        dirs = means[None, :, :] - campos[:, None, :]
        cols = spherical_harmonics(sh_degree, dirs, colors[None].expand(C, -1, -1, -1))
        cols = torch.clamp(cols + 0.5, min=0.0)
        vis = (radii > 0).all(-1)
        cols = torch.where(vis[..., None], cols, torch.zeros_like(cols))
    else:
        cols = colors if colors.dim() == 3 else colors[None].expand(C, -1, -1)
    tw = math.ceil(width / tile_size)
    th = math.ceil(height / tile_size)
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size, tw, th)
    isect_offsets = isect_offset_encode(isect_ids, C, tw, th)
    if absgrad:
        means2d.retain_grad() if means2d.requires_grad else None
    render, alphas, last_ids = rasterize_to_pixels(
        means2d, conics, cols, op, width, height, tile_size, isect_offsets, flatten_ids,
        backgrounds=backgrounds, marginal=marginal)
    meta = dict(radii=radii, means2d=means2d, depths=depths, conics=conics, opacities=op,
                colors=cols, tiles_per_gauss=tiles_per_gauss, isect_ids=isect_ids,
                flatten_ids=flatten_ids, isect_offsets=isect_offsets, last_ids=last_ids,
                tile_width=tw, tile_height=th, tile_size=tile_size, width=width, height=height,
                n_cameras=C)
    return render, alphas, meta


# -------------------------------------------------------------------------------- loss
def _gauss_window(size: int = 11, sigma: float = 1.5, dtype=torch.float64) -> torch.Tensor:
    xs = torch.arange(size, dtype=dtype) - size // 2
    g = torch.exp(-(xs ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def ssim(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """Mean SSIM, 11x11 Gaussian window sigma 1.5, zero 'same' padding, C1=.01^2 C2=.03^2
    (the original 3DGS `ssim()` / fused-ssim 'same' mode).  img[B,3,H,W] in [0,1]."""
    ch = img1.shape[1]
    g = _gauss_window(dtype=img1.dtype)
    w2d = (g[:, None] * g[None, :])[None, None].expand(ch, 1, 11, 11).contiguous()

    def blur(x):
        return F.conv2d(x, w2d, padding=5, groups=ch)

    mu1, mu2 = blur(img1), blur(img2)
    s11 = blur(img1 * img1) - mu1 * mu1
    s22 = blur(img2 * img2) - mu2 * mu2
    s12 = blur(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean()


def photometric_loss(render: torch.Tensor, gt: torch.Tensor, ssim_lambda: float = 0.2):
    """(1-l)*L1 + l*(1-SSIM); render/gt are [C,H,W,3]."""
    l1 = (render - gt).abs().mean()
    s = ssim(render.permute(0, 3, 1, 2), gt.permute(0, 3, 1, 2))
    return (1.0 - ssim_lambda) * l1 + ssim_lambda * (1.0 - s)


def scale_regularisation(scales_exp: torch.Tensor, max_ratio: float = 10.0, weight: float = 0.1):
    """splatfacto `use_scale_regularization` (reference main.py:1288): PhysGaussian ratio term."""
    ratio = scales_exp.amax(-1) / scales_exp.amin(-1)
    return weight * (torch.clamp(ratio, min=max_ratio) - max_ratio).mean()


# ------------------------------------------------------------------------------- Adam
def adam_step(p, g, m, v, step: int, lr: float, b1: float = 0.9, b2: float = 0.999,
              eps: float = 1e-15):
    """torch.optim.Adam semantics (no weight decay, no amsgrad); returns new (p, m, v)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


# -------------------------------------------------------------- densify / prune (ADC)
def strategy_update_state(state: Dict, means2d_grad: torch.Tensor, radii: torch.Tensor,
                          width: int, height: int, n_cameras: int):
    """gsplat DefaultStrategy._update_state: accumulate screen-space gradient norms."""
    g = means2d_grad.clone()
    g[..., 0] *= width / 2.0 * n_cameras
    g[..., 1] *= height / 2.0 * n_cameras
    sel = (radii > 0).all(-1)
    norms = g.norm(dim=-1)
    state["grad2d"] = state["grad2d"] + torch.where(sel, norms, torch.zeros_like(norms)).sum(0)
    state["count"] = state["count"] + sel.to(norms.dtype).sum(0)
    return state


def strategy_masks(state: Dict, scales_exp: torch.Tensor, opacities_sig: torch.Tensor, step: int,
                   scene_scale: float = 1.0, prune_opa: float = 0.005, grow_grad2d: float = 0.0002,
                   grow_scale3d: float = 0.01, prune_scale3d: float = 0.1, reset_every: int = 3000,
                   grow_scale2d: float = 0.05, prune_scale2d: float = 0.15, refine_scale2d_stop_iter: int = 0):
    """Decision masks of DefaultStrategy._grow_gs / _prune_gs (before any RNG is drawn).  is_prune is the mask over the
    ORIGINAL Gaussians, before growth; upstream evaluates it on the grown set, where a split's samples are 1.6 x smaller and
    every child carries its parent's radius statistic (the caller applies the 1.6).
    Screen-size rules (refine_scale2d_stop_iter > 0; nerfstudio splatfacto: 0.05 / 0.15 / 4000, reference main.py:1270-1306):
    state["radii"] = running max of radius / max(W, H); while step < the stop iteration, split also where it exceeds
    grow_scale2d (a Gaussian can then be BOTH duplicated and split: duplicate() runs first, split() replaces the original),
    and once step > reset_every prune also where it exceeds prune_scale2d."""
    grads = state["grad2d"] / state["count"].clamp(min=1)
    high = grads > grow_grad2d
    small = scales_exp.amax(-1) <= grow_scale3d * scene_scale
    is_dupli = high & small
    is_split = high & ~small
    screen = step < refine_scale2d_stop_iter
    if screen:
        is_split = is_split | (state["radii"] > grow_scale2d)
    is_prune = opacities_sig < prune_opa
    if step > reset_every:
        is_prune = is_prune | (scales_exp.amax(-1) > prune_scale3d * scene_scale)
        if screen:
            is_prune = is_prune | (state["radii"] > prune_scale2d)
    return is_dupli, is_split, is_prune


# ------------------------------------------------------- exact contributing (tile, splat) set
def contributing_pairs(means2d, conics, opacities, radii, width: int, height: int, tile_size: int = 16):
    """Brute force, float64: the set of (camera, tile, Gaussian) triples in which at least one
    pixel centre of the tile receives alpha = o exp(-sigma) >= 1/255 from the Gaussian.  Any
    binning scheme has to keep at least these; gsplat's bounding-box binning keeps more.
    Returns a sorted int64 tensor of codes ((cam * n_tiles + tile) * N + gaussian)."""
    C, N = opacities.shape
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float64) + 0.5,
                            torch.arange(width, dtype=torch.float64) + 0.5, indexing="ij")
    tile_of_px = ((ys.long() // tile_size) * tw + (xs.long() // tile_size)).flatten()
    out = []
    for c in range(C):
        for g in torch.nonzero((radii[c] > 0).all(-1)).flatten().tolist():
            dx = means2d[c, g, 0].double() - xs.flatten()
            dy = means2d[c, g, 1].double() - ys.flatten()
            A, B, Cc = conics[c, g].double()
            sigma = 0.5 * (A * dx * dx + Cc * dy * dy) + B * dx * dy
            hit = (sigma >= 0) & (opacities[c, g].double() * torch.exp(-sigma) >= ALPHA_THRESHOLD)
            tiles = torch.unique(tile_of_px[hit])
            out.append((c * tw * th + tiles) * N + g)
    return torch.sort(torch.cat(out)).values if out else torch.zeros(0, dtype=torch.int64)


# ------------------------------------------------------------------------ MCMC relocation
def mcmc_relocation(opacities: torch.Tensor, scales: torch.Tensor, ratios: torch.Tensor):
    """gsplat `compute_relocation` ("3DGS as MCMC", eq. 9): a Gaussian replaced by `ratio` co-located
    copies: o' = 1 - (1-o)^(1/ratio); s' = s o / sum_{i=1..ratio} sum_{k<i} C(i-1,k) (-1)^k o'^(k+1) / sqrt(k+1)."""
    no = 1.0 - (1.0 - opacities) ** (1.0 / ratios.to(opacities.dtype))
    denom = torch.zeros_like(opacities)
    for j in range(opacities.shape[0]):
        d = 0.0
        for i in range(1, int(ratios[j]) + 1):
            for k in range(i):
                d += math.comb(i - 1, k) * (-1) ** k * float(no[j]) ** (k + 1) / math.sqrt(k + 1)
        denom[j] = d
    return no, scales * (opacities / denom)[:, None]
