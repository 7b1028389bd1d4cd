"""Tensor-level wrappers over the mi3dgs C-ABI, and the gsplat-shaped `rasterization()`.

The reference reaches this operator surface only through subprocesses
(source/container/src/main.py:1312 `ns-train splatfacto`, main.py:1343
`gsplat/examples/simple_trainer.py`); the names, argument meaning and error behaviour below
mirror upstream gsplat's `rasterization()` so the parity tests read like its tests
(SURVEY.md 8b tier 2).  PyTorch is only plumbing here: device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib

SPLAT_STRIDE = 16
GRAD_STRIDE = 16
FLAG_LOG_SCALES = 1
FLAG_LOGIT_OPAC = 2
FLAG_ANTIALIASED = 4
FLAG_CLEAR_VSPLATS = 8      # project_bwd / project_bwd_adam: clear the v_splats rows they read (include/mi3dgs.h)
FLAG_ONLY_CULLED_GROUPS = 16     # project_bwd_adam: only the 64-Gaussian groups without a visible member (pure Adam stream)
FLAG_ONLY_VISIBLE_GROUPS = 32    # project_bwd_adam: only the groups with one
COLOR_SH, COLOR_PER_GAUSSIAN, COLOR_PER_CAMERA = 0, 1, 2


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _chk(t: torch.Tensor, name: str, shape=None, dtype=torch.float32):
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (mi3dgs has no CPU path)")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if shape is not None:
        if len(shape) != t.dim() or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {shape}")


_WS: Dict[Tuple[int, str], torch.Tensor] = {}


def workspace(nbytes: int, device, tag: str = "bin") -> torch.Tensor:
    """Cached scratch buffer (grown geometrically, never shrunk) -- the C-ABI never allocates."""
    key = (torch.device(device).index or 0, tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# ------------------------------------------------------------------------------- stages
def project_fwd(means, quats, scales, opacities, viewmats, Ks, width, height, *, sh0=None, shN=None,
                colors=None, sh_degree=0, eps2d=0.3, near_plane=0.01, far_plane=1e10, radius_clip=0.0,
                flags=0, radii=None, splats=None, depth_keys=None):
    N, Cn = means.shape[0], viewmats.shape[0]
    dev = means.device
    _chk(means, "means", (N, 3)); _chk(quats, "quats", (N, 4)); _chk(scales, "scales", (N, 3))
    _chk(viewmats, "viewmats", (Cn, 4, 4)); _chk(Ks, "Ks", (Cn, 3, 3))
    if opacities is not None:
        _chk(opacities, "opacities", (N,))
    if colors is None:
        mode = COLOR_SH
        _chk(sh0, "sh0", (N, 1, 3))
        if sh_degree > 0:
            _chk(shN, "shN", (N, 15, 3))
    else:
        mode = COLOR_PER_CAMERA if colors.dim() == 3 else COLOR_PER_GAUSSIAN
        _chk(colors, "colors", (Cn, N, 3) if mode == COLOR_PER_CAMERA else (N, 3))
    if radii is None:
        radii = torch.empty(Cn, N, 2, dtype=torch.int32, device=dev)
    if splats is None:
        splats = torch.empty(Cn, N, SPLAT_STRIDE, dtype=torch.float32, device=dev)
    _lib.call("mi3dgs_project_fwd", Cn, N, _p(means), _p(quats), _p(scales), _p(opacities), _p(sh0), _p(shN),
              _p(colors), mode, int(sh_degree), _p(viewmats), _p(Ks), int(width), int(height), float(eps2d),
              float(near_plane), float(min(far_plane, 3.0e38)), float(radius_clip), int(flags), _p(radii),
              _p(splats), _p(depth_keys), _stream(dev))
    return radii, splats


def project_bwd(means, quats, scales, opacities, viewmats, Ks, width, height, radii, splats, v_splats, *,
                sh0=None, shN=None, color_mode=COLOR_SH, sh_degree=0, eps2d=0.3, flags=0, out=None,
                stats=None, stat_use_abs=False):
    """Returns dict of gradients (v_means, v_quats, v_scales, v_opacities, v_sh0/v_shN or v_colors)."""
    N, Cn = means.shape[0], viewmats.shape[0]
    dev = means.device
    o = out if out is not None else {}

    def buf(name, shape):
        t = o.get(name)
        if t is None:
            t = torch.empty(shape, dtype=torch.float32, device=dev)
            o[name] = t
        return t

    v_means, v_quats, v_scales = buf("v_means", (N, 3)), buf("v_quats", (N, 4)), buf("v_scales", (N, 3))
    v_opac = buf("v_opacities", (N,)) if opacities is not None else None
    v_sh0 = v_shN = v_colors = None
    if color_mode == COLOR_SH:
        v_sh0, v_shN = buf("v_sh0", (N, 1, 3)), buf("v_shN", (N, 15, 3))
    elif color_mode == COLOR_PER_GAUSSIAN:
        v_colors = buf("v_colors", (N, 3))
    else:
        v_colors = buf("v_colors", (Cn, N, 3))
    sg = sc = sr = None
    if stats is not None:
        sg, sc, sr = stats.get("grad2d"), stats.get("count"), stats.get("radii")
    _lib.call("mi3dgs_project_bwd", Cn, N, _p(means), _p(quats), _p(scales), _p(opacities), _p(sh0), _p(shN),
              int(color_mode), int(sh_degree), _p(viewmats), _p(Ks), int(width), int(height), float(eps2d),
              int(flags), _p(radii), _p(splats), _p(v_splats), _p(v_means), _p(v_quats), _p(v_scales), _p(v_opac),
              _p(v_sh0), _p(v_shN), _p(v_colors), _p(sg), _p(sc), _p(sr), int(bool(stat_use_abs)), _stream(dev))
    return o


def project_bwd_adam(params, exp_avg, exp_avg_sq, lrs, step, viewmat, K, width, height, radii, splats, v_splats, *,
                     n, sh_degree, eps2d=0.3, flags=0, beta1=0.9, beta2=0.999, eps=1e-15, scale_reg_weight=0.0,
                     scale_reg_max_ratio=10.0, stats=None, stat_use_abs=False, mcmc_opacity_reg=0.0, mcmc_scale_reg=0.0):
    """Single-camera backward with the Adam step fused in: `params` (list of the 6 group
    tensors means, quats, scales, opacities, sh0, shN) are updated in place, no gradients are
    materialised.  `n` = live Gaussians (the tensors may be larger: capacity).  mcmc_*: gsplat's MCMC regularisers folded in
    (mi3dgs_project_bwd_adam_mcmc; no scale_reg_weight / stats with them)."""
    dev = params[0].device
    for t in list(params) + list(exp_avg) + list(exp_avg_sq):
        _chk(t, "parameter / moment buffer")
    sg = sc = sr = None
    if stats is not None:
        sg, sc, sr = stats.get("grad2d"), stats.get("count"), stats.get("radii")
    lr = (C.c_float * 6)(*[float(x) for x in lrs])
    if mcmc_opacity_reg or mcmc_scale_reg:
        if scale_reg_weight or stats is not None:
            raise ValueError("project_bwd_adam: the MCMC regularisers go without scale_reg_weight / stats")
        _lib.call("mi3dgs_project_bwd_adam_mcmc", int(n), _p(params[0]), _p(params[1]), _p(params[2]), _p(params[3]),
                  _p(params[4]), _p(params[5]), int(sh_degree), _p(viewmat), _p(K), int(width), int(height), float(eps2d),
                  int(flags), _p(radii), _p(splats), _p(v_splats), _ptr_array(exp_avg), _ptr_array(exp_avg_sq), lr, int(step),
                  float(beta1), float(beta2), float(eps), float(mcmc_opacity_reg), float(mcmc_scale_reg), _stream(dev))
        return
    _lib.call("mi3dgs_project_bwd_adam", int(n), _p(params[0]), _p(params[1]), _p(params[2]), _p(params[3]),
              _p(params[4]), _p(params[5]), int(sh_degree), _p(viewmat), _p(K), int(width), int(height), float(eps2d),
              int(flags), _p(radii), _p(splats), _p(v_splats), _ptr_array(exp_avg), _ptr_array(exp_avg_sq), lr, int(step),
              float(beta1), float(beta2), float(eps), float(scale_reg_weight), float(scale_reg_max_ratio), _p(sg), _p(sc),
              _p(sr), int(bool(stat_use_abs)), _stream(dev))


def adam_culled_groups(params, exp_avg, exp_avg_sq, lrs, step, radii, *, n, beta1=0.9, beta2=0.999, eps=1e-15,
                       scale_reg_weight=0.0, scale_reg_max_ratio=10.0):
    """Adam on the aligned 64-Gaussian groups none of whose members is visible in `radii` (zero gradient: a pure stream).
    The counterpart of project_bwd_adam(flags | FLAG_ONLY_VISIBLE_GROUPS)."""
    dev = params[0].device
    lr = (C.c_float * 6)(*[float(x) for x in lrs])
    _lib.call("mi3dgs_adam_culled_groups", int(n), _ptr_array(params), _ptr_array(exp_avg), _ptr_array(exp_avg_sq), _p(radii), lr,
              int(step), float(beta1), float(beta2), float(eps), float(scale_reg_weight), float(scale_reg_max_ratio), _stream(dev))


def bin_tiles(radii, splats, width, height, tile_size=16, *, max_isect: Optional[int] = None,
              want_isect_ids: bool = False, want_tiles_per_gauss: bool = False, tight: bool = False,
              fused: bool = True, depth_keys: Optional[torch.Tensor] = None, radii_in_records: bool = False,
              want_tile_keys: bool = True):
    """Tile binning.  tight=False reproduces gsplat's bounding-box tile lists; tight=True drops
    the (tile, splat) pairs the ellipse sigma <= ln(255 o) cannot reach (identical renders and
    gradients, fewer intersections).  With max_isect=None the intersection count is read back (one host
    sync) and the outputs are sized exactly; otherwise outputs hold max_isect entries and
    the live count stays on the device (no sync).  depth_keys [C,N] int32 (fused path only): the sort
    keys mi3dgs_project_fwd wrote beside its records; they are consumed.  radii_in_records: the records come from
    mi3dgs_project_fwd (radii in slots 11, 12): one gather per splat in the emit pass.  want_tile_keys=False (fused
    path): the sorted tile keys are not returned and their buffer is scratch, which lets the library sort 16-bit keys
    when every tile id fits (the training step and the renderer only use flatten_ids and isect_offsets)."""
    tight = int(bool(tight)) | (2 if radii_in_records else 0) | (0 if want_tile_keys else 4)
    Cn, N = radii.shape[0], radii.shape[1]
    dev = radii.device
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    # (both entry points clear the count themselves; only a call that returns early on an empty scene relies on this fill)
    n_isect = torch.zeros(1, dtype=torch.int32, device=dev) if (N == 0 or max_isect == 0) else torch.empty(1, dtype=torch.int32, device=dev)
    tpg = torch.empty(Cn, N, dtype=torch.int32, device=dev) if want_tiles_per_gauss else None
    # the phase-1 layout depends on max_isect only for its tail, so count with cap 0 when unknown
    cap_known = max_isect is not None
    cap = int(max_isect) if cap_known else 0
    ws_bytes = _lib.lib().mi3dgs_bin_workspace_bytes(Cn, N, cap)
    ws = workspace(ws_bytes, dev)
    st = _stream(dev)
    if cap_known and fused:
        # capacity given: one call, counting and emission fused in a chained pass
        flatten_ids = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
        tile_keys = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
        offsets = torch.empty(Cn, th, tw, dtype=torch.int32, device=dev)
        isect_ids = torch.empty(max(cap, 1), dtype=torch.int64, device=dev) if want_isect_ids else None
        _lib.call("mi3dgs_bin_tiles", Cn, N, _p(radii), _p(splats), tile_size, tw, th, int(height), int(tight),
                  _p(n_isect), cap, _p(flatten_ids), _p(tile_keys), _p(offsets), _p(isect_ids), _p(tpg), _p(depth_keys),
                  _p(ws), ws.numel(), st)
        out = dict(n_isect=n_isect, flatten_ids=flatten_ids, tile_keys=tile_keys if want_tile_keys else None,
                   isect_offsets=offsets, tile_width=tw, tile_height=th, max_isect=cap)
        if want_isect_ids:
            out["isect_ids"] = isect_ids
        if want_tiles_per_gauss:
            out["tiles_per_gauss"] = tpg
        return out
    _lib.call("mi3dgs_bin_count", Cn, N, _p(radii), _p(splats), tile_size, tw, th, int(height), int(tight),
              _p(tpg), _p(n_isect), _p(ws), ws.numel(), cap, st)
    if not cap_known:
        cap = int(n_isect.item())
        ws_bytes = _lib.lib().mi3dgs_bin_workspace_bytes(Cn, N, cap)
        if ws.numel() < ws_bytes:
            # growing would lose phase-1 state: keep the old buffer's head by copying it over
            old = ws
            ws = torch.empty(int(ws_bytes * 1.25) + 256, dtype=torch.uint8, device=dev)
            ws[: old.numel()].copy_(old)
            _WS[(torch.device(dev).index or 0, "bin")] = ws
    flatten_ids = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
    tile_keys = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
    offsets = torch.empty(Cn, th, tw, dtype=torch.int32, device=dev)
    isect_ids = torch.empty(max(cap, 1), dtype=torch.int64, device=dev) if want_isect_ids else None
    _lib.call("mi3dgs_bin_emit", Cn, N, _p(radii), _p(splats), tile_size, tw, th, int(height), int(tight),
              _p(n_isect), cap, _p(flatten_ids),
              _p(tile_keys), _p(offsets), _p(isect_ids), _p(ws), ws.numel(), st)
    out = dict(n_isect=n_isect, flatten_ids=flatten_ids[:cap] if not cap_known else flatten_ids,
               tile_keys=tile_keys[:cap] if not cap_known else tile_keys, isect_offsets=offsets,
               tile_width=tw, tile_height=th, max_isect=cap)
    if want_isect_ids:
        out["isect_ids"] = isect_ids[:cap] if not cap_known else isect_ids
    if want_tiles_per_gauss:
        out["tiles_per_gauss"] = tpg
    return out


def set_raster_fwd_segments(on: bool) -> None:
    """Process-wide: with a segment workspace, rasterize_fwd walks tile lists of more than 256 entries as segments side by
    side (on, the default) or serially, leaving checkpoints for the backward only (off)."""
    _lib.call("mi3dgs_debug_set_raster_fwd_segments", int(bool(on)))


def raster_seg_workspace(binning, Cn, device, out=None):
    """The workspace through which rasterize_fwd hands per-pixel checkpoints to rasterize_bwd (include/mi3dgs.h, "Segment
    workspace"): 4 KB per possible 512-entry boundary, touched only where a tile's list really is that long."""
    n_tiles = Cn * binning["tile_width"] * binning["tile_height"]
    nbytes = int(_lib.lib().mi3dgs_raster_seg_workspace_bytes(n_tiles, int(binning["max_isect"])))
    ws = out.get("seg_ws") if out is not None else None
    if ws is None or ws.numel() < nbytes or ws.device != device:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        # control words cleared once; every rasterize_bwd given the workspace leaves them clear for the next forward
        _lib.call("mi3dgs_raster_seg_workspace_init", _p(ws), ws.numel(), _stream(device))
        if out is not None:
            out["seg_ws"] = ws
    # exactly the size of THIS call's lists: both rasterisers derive the layout (and whether lists are long enough everywhere
    # for segments to be pointless) from the byte count
    return ws[:nbytes]


def rasterize_fwd(splats, binning, width, height, tile_size=16, backgrounds=None, out=None, seg_ws=None):
    """seg_ws: a raster_seg_workspace() tensor; hand the same one (and `render`) to rasterize_bwd of this step."""
    Cn = splats.shape[0]
    dev = splats.device
    o = out if out is not None else {}
    render = o.get("render")
    if render is None:
        render = torch.empty(Cn, height, width, 3, dtype=torch.float32, device=dev)
        alphas = torch.empty(Cn, height, width, 1, dtype=torch.float32, device=dev)
        last_ids = torch.empty(Cn, height, width, dtype=torch.int32, device=dev)
        o.update(render=render, alphas=alphas, last_ids=last_ids)
    if backgrounds is not None:
        _chk(backgrounds, "backgrounds", (Cn, 3))
    _lib.call("mi3dgs_rasterize_fwd", Cn, int(width), int(height), tile_size, binning["tile_width"],
              binning["tile_height"], _p(splats), _p(binning["isect_offsets"]), _p(binning["flatten_ids"]),
              _p(binning["n_isect"]), _p(backgrounds), _p(o["render"]), _p(o["alphas"]), _p(o["last_ids"]),
              _p(seg_ws), 0 if seg_ws is None else seg_ws.numel(), _stream(dev))
    return o["render"], o["alphas"], o["last_ids"]


def rasterize_bwd(splats, binning, width, height, alphas, last_ids, v_render, v_alphas, tile_size=16,
                  backgrounds=None, absgrad=False, v_splats=None, render=None, seg_ws=None):
    Cn, N = splats.shape[0], splats.shape[1]
    dev = splats.device
    if v_splats is None:
        v_splats = torch.zeros(Cn, N, GRAD_STRIDE, dtype=torch.float32, device=dev)
    _chk(v_render, "v_render", (Cn, height, width, 3)); _chk(v_alphas, "v_alphas", (Cn, height, width, 1))
    if seg_ws is not None:
        if render is None:
            raise ValueError("rasterize_bwd: seg_ws needs the forward's render")
        _chk(render, "render", (Cn, height, width, 3))
    _lib.call("mi3dgs_rasterize_bwd", Cn, int(width), int(height), tile_size, binning["tile_width"],
              binning["tile_height"], _p(splats), _p(binning["isect_offsets"]), _p(binning["flatten_ids"]),
              _p(binning["n_isect"]), _p(backgrounds), _p(alphas), _p(last_ids), _p(v_render), _p(v_alphas),
              int(bool(absgrad)), _p(v_splats), int(N), _p(render if seg_ws is not None else None), _p(seg_ws),
              0 if seg_ws is None else seg_ws.numel(), _stream(dev))
    return v_splats


def loss_fwd(render, target, scratch=None, want_sums=True):
    """Returns (sums[2] device tensor = {sum|r-t|, sum SSIM}, scratch dict for loss_bwd).  want_sums=False: the sums are neither
    cleared nor written (None is returned for them): a training step that does not report its loss saves the clear's launch."""
    Cn, H, W, _ = render.shape
    dev = render.device
    _chk(render, "render", (Cn, H, W, 3))
    _chk(target, "target", (Cn, H, W, 3), dtype=torch.uint8 if target.dtype == torch.uint8 else torch.float32)
    s = scratch if scratch is not None else {}
    if "dm1" not in s:
        s["dm1"] = torch.empty_like(render); s["dm2"] = torch.empty_like(render); s["dm3"] = torch.empty_like(render)
        s["sums"] = torch.zeros(2, dtype=torch.float32, device=dev)
    sums = s["sums"] if want_sums else None
    if want_sums:
        sums.zero_()
    if target.dtype == torch.uint8:         # straight from the uint8 image cache: value / 255 formed inside the kernel
        _lib.call("mi3dgs_loss_fwd_u8", Cn, H, W, _p(render), _p(target), 1.0 / 255.0, _p(s["dm1"]), _p(s["dm2"]), _p(s["dm3"]),
                  _p(sums), _stream(dev))
    else:
        _lib.call("mi3dgs_loss_fwd", Cn, H, W, _p(render), _p(target), _p(s["dm1"]), _p(s["dm2"]), _p(s["dm3"]),
                  _p(sums), _stream(dev))
    return sums, s


def loss_bwd(render, target, scratch, ssim_lambda=0.2, loss_scale=1.0, v_render=None):
    Cn, H, W, _ = render.shape
    if v_render is None:
        v_render = torch.empty_like(render)
    if target.dtype == torch.uint8:
        _lib.call("mi3dgs_loss_bwd_u8", Cn, H, W, _p(render), _p(target), 1.0 / 255.0, _p(scratch["dm1"]), _p(scratch["dm2"]),
                  _p(scratch["dm3"]), float(ssim_lambda), float(loss_scale), _p(v_render), _stream(render.device))
    else:
        _lib.call("mi3dgs_loss_bwd", Cn, H, W, _p(render), _p(target), _p(scratch["dm1"]), _p(scratch["dm2"]),
                  _p(scratch["dm3"]), float(ssim_lambda), float(loss_scale), _p(v_render), _stream(render.device))
    return v_render


def loss_value(sums: torch.Tensor, numel: int, ssim_lambda: float = 0.2) -> torch.Tensor:
    return (1.0 - ssim_lambda) * sums[0] / numel + ssim_lambda * (1.0 - sums[1] / numel)


def scale_reg(scales_log, weight=0.1, max_ratio=10.0, v_scales=None, loss_sum=None):
    _lib.call("mi3dgs_scale_reg", scales_log.shape[0], _p(scales_log), float(weight), float(max_ratio),
              _p(v_scales), _p(loss_sum), _stream(scales_log.device))


def _ptr_array(ts: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr()
    return arr


def adam_step(params: Sequence[torch.Tensor], grads, exp_avg, exp_avg_sq, lrs: Sequence[float], step: int,
              beta1=0.9, beta2=0.999, eps=1e-15, numel: Optional[Sequence[int]] = None, grad_scale: float = 1.0):
    n = len(params)
    for i in range(n):
        for t, nm in ((params[i], "param"), (grads[i], "grad"), (exp_avg[i], "exp_avg"), (exp_avg_sq[i], "exp_avg_sq")):
            _chk(t, nm)
    ne = (C.c_longlong * n)(*[int(numel[i]) if numel is not None else params[i].numel() for i in range(n)])
    lr = (C.c_float * n)(*[float(x) for x in lrs])
    _lib.call("mi3dgs_adam_step", n, _ptr_array(params), _ptr_array(grads), _ptr_array(exp_avg),
              _ptr_array(exp_avg_sq), ne, lr, int(step), float(beta1), float(beta2), float(eps), float(grad_scale),
              _stream(params[0].device))


def sort_pairs_u32(keys: torch.Tensor, vals: torch.Tensor, nbits: int = 32):
    """Stable ascending in-place sort of (int32-viewed-as-u32 keys, vals)."""
    n = keys.numel()
    nbytes = _lib.lib().mi3dgs_sort_workspace_bytes(n)
    ws = workspace(nbytes, keys.device, "sort")
    _lib.call("mi3dgs_sort_pairs_u32", _p(keys), _p(vals), n, int(nbits), _p(ws), ws.numel(), _stream(keys.device))


def scan_exclusive_u32(x: torch.Tensor, out: Optional[torch.Tensor] = None, total: Optional[torch.Tensor] = None):
    n = x.numel()
    if out is None:
        out = torch.empty_like(x)
    nbytes = _lib.lib().mi3dgs_scan_workspace_bytes(n)
    ws = workspace(nbytes, x.device, "scan")
    _lib.call("mi3dgs_scan_exclusive_u32", _p(x), _p(out), n, _p(total), _p(ws), ws.numel(), _stream(x.device))
    return out


# ------------------------------------------------------------------------- input side
def knn(points: torch.Tensor, k: int = 3, want_idx: bool = False):
    """Exact k-NN of points[n,3] (float32, on the GPU), the point itself excluded: squared
    distances [n,k] ascending (+inf where fewer than k other points exist) and, on request, the
    neighbour indices [n,k] (int32, -1 where missing)."""
    _chk(points, "points", (points.shape[0], 3))
    n = points.shape[0]
    d2 = torch.empty(n, k, dtype=torch.float32, device=points.device)
    idx = torch.empty(n, k, dtype=torch.int32, device=points.device) if want_idx else None
    nbytes = _lib.lib().mi3dgs_knn_workspace_bytes(n)
    ws = workspace(nbytes, points.device, "knn")
    _lib.call("mi3dgs_knn", n, _p(points), int(k), _p(d2), _p(idx), _p(ws), ws.numel(), _stream(points.device))
    return (d2, idx) if want_idx else d2


def image_downscale_area(img_u8: torch.Tensor, out_h: int, out_w: int, as_float: bool = False) -> torch.Tensor:
    """INTER_AREA resize of an [H,W,Ch] uint8 image on the GPU -> [out_h,out_w,Ch] uint8, or float32 in [0,1]."""
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or not img_u8.is_cuda or not img_u8.is_contiguous():
        raise ValueError("image_downscale_area: expected a contiguous [H,W,Ch] uint8 tensor on the GPU")
    H, W, Ch = img_u8.shape
    out = torch.empty(out_h, out_w, Ch, dtype=torch.float32 if as_float else torch.uint8, device=img_u8.device)
    _lib.call("mi3dgs_image_downscale_area", _p(img_u8), H, W, Ch, _p(out), int(out_h), int(out_w), int(as_float),
              _stream(img_u8.device))
    return out


def image_undistort(img_u8: torch.Tensor, k_src, k_dst, dist, out_h: int, out_w: int, fisheye: bool = False,
                    as_float: bool = False) -> torch.Tensor:
    """Pinhole re-sampling of a distorted [H,W,Ch] uint8 image on the GPU (cv2.undistort semantics):
    k_src / k_dst = (fx, fy, cx, cy), dist = OpenCV order (k1 k2 p1 p2 k3 k4 k5 k6) or fisheye k1..k4."""
    import ctypes as C
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or not img_u8.is_cuda or not img_u8.is_contiguous():
        raise ValueError("image_undistort: expected a contiguous [H,W,Ch] uint8 tensor on the GPU")
    H, W, Ch = img_u8.shape
    out = torch.empty(out_h, out_w, Ch, dtype=torch.float32 if as_float else torch.uint8, device=img_u8.device)
    d = [float(x) for x in dist]
    ks, kd, dd = (C.c_float * 4)(*[float(x) for x in k_src]), (C.c_float * 4)(*[float(x) for x in k_dst]), (C.c_float * max(len(d), 1))(*d)
    _lib.call("mi3dgs_image_undistort", _p(img_u8), H, W, Ch, _p(out), int(out_h), int(out_w), ks, kd, int(bool(fisheye)), dd,
              len(d), int(as_float), _stream(img_u8.device))
    return out


def image_u8_to_f32(img_u8: torch.Tensor, out: Optional[torch.Tensor] = None, scale: float = 1.0 / 255.0):
    """uint8 image (any shape, GPU) -> float32 * scale: the device image cache keeps u8, a step reads f32."""
    if img_u8.dtype != torch.uint8 or not img_u8.is_cuda or not img_u8.is_contiguous():
        raise ValueError("image_u8_to_f32: expected a contiguous uint8 tensor on the GPU")
    if out is None:
        out = torch.empty(img_u8.shape, dtype=torch.float32, device=img_u8.device)
    _lib.call("mi3dgs_image_u8_to_f32", _p(img_u8), img_u8.numel(), _p(out), float(scale), _stream(img_u8.device))
    return out


# --------------------------------------------------------------- gsplat-shaped operator
class _Rasterization(torch.autograd.Function):
    """project -> bin -> rasterize as one differentiable op (backward = rasterize_bwd -> project_bwd)."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, sh0, shN, colors, viewmats, Ks, backgrounds, cfg):
        W, H = cfg["width"], cfg["height"]
        flags = FLAG_ANTIALIASED if cfg["antialiased"] else 0
        radii, splats = project_fwd(means, quats, scales, opacities, viewmats, Ks, W, H, sh0=sh0, shN=shN,
                                    colors=colors, sh_degree=cfg["sh_degree"], eps2d=cfg["eps2d"],
                                    near_plane=cfg["near_plane"], far_plane=cfg["far_plane"],
                                    radius_clip=cfg["radius_clip"], flags=flags)
        binning = bin_tiles(radii, splats, W, H, cfg["tile_size"], want_isect_ids=cfg.get("want_isect_ids", False),
                            want_tiles_per_gauss=cfg.get("want_isect_ids", False))
        # a backward will follow: let the forward leave checkpoints, so that long tile lists are walked in segments
        seg_ws = raster_seg_workspace(binning, splats.shape[0], splats.device) if cfg.get("segments", True) and any(
            ctx.needs_input_grad) else None
        render, alphas, last_ids = rasterize_fwd(splats, binning, W, H, cfg["tile_size"], backgrounds, seg_ws=seg_ws)
        ctx.cfg, ctx.flags, ctx.binning, ctx.seg_ws = cfg, flags, binning, seg_ws
        ctx.color_mode = COLOR_SH if colors is None else (COLOR_PER_CAMERA if colors.dim() == 3 else COLOR_PER_GAUSSIAN)
        ctx.save_for_backward(means, quats, scales, opacities, sh0, shN, viewmats, Ks, backgrounds, radii, splats,
                              alphas, last_ids, render)
        cfg["_meta"].update(radii=radii, splats=splats, last_ids=last_ids, seg_ws=seg_ws, **binning)
        ctx.mark_non_differentiable(last_ids)
        return render, alphas, last_ids

    @staticmethod
    def backward(ctx, v_render, v_alphas, _):
        (means, quats, scales, opacities, sh0, shN, viewmats, Ks, backgrounds, radii, splats, alphas,
         last_ids, render) = ctx.saved_tensors
        cfg = ctx.cfg
        W, H = cfg["width"], cfg["height"]
        v_splats = rasterize_bwd(splats, ctx.binning, W, H, alphas, last_ids, v_render.contiguous(),
                                 v_alphas.contiguous(), cfg["tile_size"], backgrounds, cfg["absgrad"],
                                 render=render, seg_ws=ctx.seg_ws)
        cfg["_meta"]["v_splats"] = v_splats
        g = project_bwd(means, quats, scales, opacities, viewmats, Ks, W, H, radii, splats, v_splats, sh0=sh0,
                        shN=shN, color_mode=ctx.color_mode, sh_degree=cfg["sh_degree"], eps2d=cfg["eps2d"],
                        flags=ctx.flags)
        return (g["v_means"], g["v_quats"], g["v_scales"], g.get("v_opacities"), g.get("v_sh0"), g.get("v_shN"),
                g.get("v_colors"), None, None, None, None)


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width: int, height: int,
                  near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0, eps2d: float = 0.3,
                  sh_degree: Optional[int] = None, packed: bool = False, tile_size: int = 16,
                  backgrounds: Optional[torch.Tensor] = None, render_mode: str = "RGB", sparse_grad: bool = False,
                  absgrad: bool = False, rasterize_mode: str = "classic", channel_chunk: int = 32,
                  want_isect_ids: bool = False, segments: bool = True):
    """Drop-in for gsplat `rasterization()` (same argument names and meaning).

    means[N,3], quats[N,4] (wxyz, any norm), scales[N,3] (exp'd), opacities[N] (sigmoid'd),
    colors[N,K,3] SH coefficients when sh_degree is given else [N,3] / [C,N,3],
    viewmats[C,4,4], Ks[C,3,3]  ->  render[C,H,W,3], alphas[C,H,W,1], meta.
    `packed` is accepted and ignored (the result is identical; this engine always streams the
    dense [C,N] records).  Unsupported upstream options raise, they are never ignored silently.
    """
    if render_mode != "RGB":
        raise NotImplementedError("mi3dgs.rasterization: only render_mode='RGB' is implemented")
    if sparse_grad:
        raise NotImplementedError("mi3dgs.rasterization: sparse_grad is not implemented")
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"rasterize_mode must be 'classic' or 'antialiased', got {rasterize_mode!r}")
    if tile_size != 16:
        raise NotImplementedError("mi3dgs.rasterization: tile_size must be 16")
    N = means.shape[0]
    if quats.shape != (N, 4) or scales.shape != (N, 3) or opacities.shape != (N,):
        raise ValueError("quats/scales/opacities must be [N,4]/[N,3]/[N]")
    sh0 = shN = cols = None
    if sh_degree is not None:
        if colors.dim() != 3 or colors.shape[2] != 3 or colors.shape[1] < (sh_degree + 1) ** 2:
            raise ValueError("with sh_degree, colors must be [N,K,3] with K >= (sh_degree+1)^2")
        sh0 = colors[:, :1, :].contiguous()
        shN = colors[:, 1:, :]
        if shN.shape[1] < 15:
            shN = torch.cat([shN, shN.new_zeros(N, 15 - shN.shape[1], 3)], dim=1)
        shN = shN.contiguous()
    else:
        if colors.shape[-1] != 3:
            raise NotImplementedError("mi3dgs.rasterization: only 3 colour channels are implemented")
        cols = colors.contiguous()
    meta: Dict = {}
    cfg = dict(width=int(width), height=int(height), tile_size=tile_size, sh_degree=int(sh_degree or 0),
               eps2d=eps2d, near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip,
               antialiased=rasterize_mode == "antialiased", absgrad=absgrad, want_isect_ids=want_isect_ids, segments=segments,
               _meta=meta)
    render, alphas, _ = _Rasterization.apply(means.contiguous(), quats.contiguous(), scales.contiguous(),
                                             opacities.contiguous(), sh0, shN, cols, viewmats.contiguous(),
                                             Ks.contiguous(), backgrounds, cfg)
    sp = meta["splats"]
    meta.update(means2d=sp[..., 0:2], conics=sp[..., 2:5], opacities=sp[..., 5], colors=sp[..., 6:9],
                depths=sp[..., 9], width=width, height=height, tile_size=tile_size, n_cameras=viewmats.shape[0])
    return render, alphas, meta
