"""3DGS training loop over the mi3dgs C-ABI: the work `Train-Stage1` does in the reference.

Mirrors what the reference launches as a subprocess --
  source/container/src/main.py:1270-1316  `ns-train splatfacto ...`        (single GPU)
  source/container/src/main.py:1318-1347  `simple_trainer.py default ...`  (multi GPU)
-- with upstream gsplat `simple_trainer.py` / DefaultStrategy defaults (SURVEY.md
Appendix A; [UPSTREAM-UNVERIFIED], the reference passes almost no hyper-parameters).

No autograd in the loop: forward, loss, backward, Adam and densify are explicit calls into
pre-allocated buffers, so one iteration is a fixed sequence of kernel launches on one stream.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from . import ops

GROUPS = ("means", "quats", "scales", "opacities", "sh0", "shN")
WIDTHS = (3, 4, 3, 1, 3, 45)
_SHAPES = {"means": (3,), "quats": (4,), "scales": (3,), "opacities": (), "sh0": (1, 3), "shN": (15, 3)}


@dataclass
class TrainConfig:
    max_steps: int = 30_000
    sh_degree: int = 3
    sh_degree_interval: int = 1000
    ssim_lambda: float = 0.2
    # Adam (gsplat simple_trainer defaults; means lr is multiplied by scene_scale and decays to 1%)
    lr_means: float = 1.6e-4
    lr_means_final_ratio: float = 0.01
    lr_scales: float = 5e-3
    lr_quats: float = 1e-3
    lr_opacities: float = 5e-2
    lr_sh0: float = 2.5e-3
    lr_shN: float = 2.5e-3 / 20
    adam_eps: float = 1e-15
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    scene_scale: float = 1.0
    # DefaultStrategy
    densify: bool = True
    prune_opa: float = 0.005
    grow_grad2d: float = 0.0002
    grow_scale3d: float = 0.01
    prune_scale3d: float = 0.1
    refine_start_iter: int = 500
    refine_stop_iter: int = 15_000
    reset_every: int = 3000
    refine_every: int = 100
    pause_refine_after_reset: int = 0
    # screen-size rules (gsplat DefaultStrategy grow_scale2d / prune_scale2d / refine_scale2d_stop_iter; 0 = off, gsplat's default.
    # nerfstudio splatfacto -- the reference's default job, main.py:1270-1306 -- sets split_screen_size 0.05, cull_screen_size
    # 0.15, stop_screen_size_at 4000): while step < the stop iteration a Gaussian whose largest radius / max(W, H) since the last
    # refine exceeds grow_scale2d is split, and (once step > reset_every) one beyond prune_scale2d is pruned
    grow_scale2d: float = 0.05
    prune_scale2d: float = 0.15
    refine_scale2d_stop_iter: int = 0
    absgrad: bool = False
    # splatfacto extras the reference turns on (main.py:1288)
    use_scale_regularization: bool = False
    scale_reg_weight: float = 0.1
    max_gauss_ratio: float = 10.0
    scale_reg_every: int = 10
    random_background: bool = False
    # splatfacto's coarse-to-fine schedule: train at 1/2^num_downscales of the resolution, doubling
    # every `resolution_schedule` steps (nerfstudio defaults 2 and 3000; gsplat's trainer: 0)
    num_downscales: int = 0
    resolution_schedule: int = 3000
    antialiased: bool = False
    near_plane: float = 0.01
    far_plane: float = 1e10
    # capacities: Gaussians (densify target buffers) and tile intersections (None = read the
    # count back every step and allocate exactly; an int = no host sync in the step)
    capacity: Optional[int] = None
    max_isect: Optional[int] = None
    # max_isect is None: size the intersection buffers from measured counts (every training view once at start and at a
    # resolution change, a device-side running peak afterwards) and grow them on demand, so that a training step never reads
    # the count back: no host sync per step, and the fused chained binning instead of the count / emit pair
    auto_isect_capacity: bool = False
    # exact ellipse-tile culling at binning time (identical renders/gradients, fewer intersections)
    tight_tiles: bool = True
    # capacity mode only: count and emit fused in one chained pass (mi3dgs_bin_tiles)
    fused_binning: bool = True
    # the backward walks tile lists longer than 512 entries in segments, from checkpoints the forward leaves (include/mi3dgs.h)
    raster_segments: bool = True
    # ... and the forward walks lists longer than 256 entries as segments side by side (process-wide switch of the library).
    # Pays where the lists are walked to their ends (renders of an MCMC-trained model: 555 -> 283 us); in training the serial walk's
    # stop at saturation is worth more (-4 .. -11 % of the step rate with this on, docs/FINDINGS_r03.md 4.2)
    raster_fwd_segments: bool = False
    # order the initial Gaussians along a Morton curve of their positions (a permutation: same training up to float summation
    # order).  Neighbours in memory are then neighbours in space: a wave of the projection kernels is culled or visible as a
    # whole, and the rasterisers' record gathers hit the cache.  Refinement keeps children next to their parents, so the order
    # survives.  Off by default (callers that compare parameters by index); the CLI and bench.py switch it on.
    spatial_sort_init: bool = False
    # fused single-GPU path: the Adam update of the 64-Gaussian groups none of whose members is visible needs no gradient; it is
    # launched on a second stream right after the binning and streams under the rasterisers (which leave HBM idle) instead of
    # after them.  Same update for every Gaussian, exactly once; pays where many groups are culled as a whole, i.e. with the
    # Gaussians in Morton order (spatial_sort_init).  "after_project" / "after_binning" / "after_raster_fwd": where the side launch is issued.
    overlap_culled_adam: Optional[str] = None
    overlap_min_gaussians: int = 100_000       # (below this a step is launch-bound and the side launch costs more than it hides)
    # Adam fused into the backward (single-GPU path; the data-parallel trainer needs the
    # gradients for its all-reduce and switches this off)
    fuse_adam: bool = True
    seed: int = 0


class GaussianModel:
    """Parameter + Adam-moment store with spare capacity and two banks (densify ping-pong)."""

    def __init__(self, params: Dict[str, torch.Tensor], capacity: Optional[int] = None, flat: bool = False, align: int = 1):
        """flat: parameters, both moments and the gradients each live in ONE allocation, group after group
        ([means | quats | scales | opacities | sh0 | shN], each [capacity, width]); the sharded optimiser
        reduce-scatters / all-gathers those buffers whole (parallel.py).  align: capacity is rounded up to it."""
        dev = params["means"].device
        self.device = dev
        self.n = int(params["means"].shape[0])
        self.capacity = int(capacity or self.n)
        if self.capacity < self.n:
            raise ValueError("capacity smaller than the initial number of Gaussians")
        if flat:
            align = max(align, 4) // 4 * 4 if align % 4 == 0 else align * 4      # every group then starts 16-byte aligned
        self.capacity = (self.capacity + align - 1) // align * align
        cap = self.capacity
        self.flat: Optional[Dict] = None
        self.banks = []
        if flat:
            tot = sum(WIDTHS) * cap
            self.flat = {k: [torch.zeros(tot, dtype=torch.float32, device=dev) for _ in range(2)] for k in ("p", "m", "v")}
            self.flat["g"] = torch.zeros(tot, dtype=torch.float32, device=dev)
            offs = [sum(WIDTHS[:i]) * cap for i in range(len(WIDTHS))]
            for b in range(2):
                self.banks.append({g: {k: self.flat[k][b][o: o + w * cap].view(cap, w) for k in ("p", "m", "v")}
                                   for g, w, o in zip(GROUPS, WIDTHS, offs)})
            self.grads = {g: self.flat["g"][o: o + w * cap].view(cap, w) for g, w, o in zip(GROUPS, WIDTHS, offs)}
        else:
            for b in range(2):
                bank = {}
                for g, w in zip(GROUPS, WIDTHS):
                    bank[g] = {k: torch.zeros(cap, w, dtype=torch.float32, device=dev) for k in ("p", "m", "v")}
                self.banks.append(bank)
            self.grads = {g: torch.zeros(cap, w, dtype=torch.float32, device=dev) for g, w in zip(GROUPS, WIDTHS)}
        self.cur = 0
        for g, w in zip(GROUPS, WIDTHS):
            self.banks[0][g]["p"][: self.n] = params[g].reshape(self.n, w).to(torch.float32)

    def _view(self, t: torch.Tensor, g: str) -> torch.Tensor:
        return t[: self.n].view((self.n,) + _SHAPES[g])

    def p(self, g: str) -> torch.Tensor:
        return self._view(self.banks[self.cur][g]["p"], g)

    def grad(self, g: str) -> torch.Tensor:
        return self._view(self.grads[g], g)

    def state(self, g: str, k: str) -> torch.Tensor:
        return self.banks[self.cur][g][k][: self.n]

    def splats_state_dict(self) -> Dict[str, torch.Tensor]:
        """The checkpoint schema the reference's exporter reads (gsplat_pt_to_ply.py:45-73)."""
        return {g: self.p(g).detach().clone() for g in GROUPS}


def morton_order(means: torch.Tensor) -> torch.Tensor:
    """Indices that order [N,3] positions along a 30-bit Morton curve of their bounding box (ties keep their order)."""
    m = means.detach().reshape(-1, 3).double()
    lo, hi = m.min(0).values, m.max(0).values
    q = ((m - lo) / (hi - lo).clamp(min=1e-30) * 1023.999).long().clamp(0, 1023)

    def spread(x):          # 10 bits -> every third bit
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        return (x | (x << 2)) & 0x09249249

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.sort(code, stable=True).indices


class Trainer:
    def __init__(self, params: Dict[str, torch.Tensor], viewmats: torch.Tensor, Ks: torch.Tensor,
                 images: torch.Tensor, width: int, height: int, cfg: Optional[TrainConfig] = None):
        self.cfg = cfg or TrainConfig()
        ops.set_raster_fwd_segments(self.cfg.raster_fwd_segments)
        if self.cfg.spatial_sort_init and params["means"].shape[0] > 1:
            perm = morton_order(params["means"])
            params = {k: v[perm.to(v.device)] for k, v in params.items()}
        self.model = GaussianModel(params, self.cfg.capacity, **self._model_layout())
        dev = self.model.device
        self.device = dev
        self.viewmats, self.Ks = viewmats.to(dev).contiguous(), Ks.to(dev).contiguous()
        # [V,H,W,3] on the device: float32 in [0,1], or the uint8 image cache (a quarter of the HBM; the loss kernels read it as it is)
        self.images = images
        self.W, self.H = int(width), int(height)
        self.W0, self.H0, self.Ks0, self._cur_d = self.W, self.H, self.Ks, 1
        self._auto_cap: Optional[int] = None       # auto_isect_capacity: current capacity, None = measure first
        self._auto_caps: Dict[int, int] = {}       # ... per resolution divisor: a render() at full size must not cost the training
        self._auto_cap_n: Dict[int, int] = {}      #     resolution its calibration (and the Gaussian count it was made at)
        self._last_cap_check_step = 0
        self.log = None                            # callable(str): where absorbed capacity overflows are reported (cli: rank 0)
        self._isect_peak = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.isect_overflows = 0
        self.step_count = 0
        cap = self.model.capacity
        self.radii = torch.empty(1, cap, 2, dtype=torch.int32, device=dev)
        self.splats = torch.empty(1, cap, ops.SPLAT_STRIDE, dtype=torch.float32, device=dev)
        self.depth_keys = torch.empty(1, cap, dtype=torch.int32, device=dev)
        self.v_splats = torch.zeros(1, cap, ops.GRAD_STRIDE, dtype=torch.float32, device=dev)
        self.raster_out: Dict = {}
        self.loss_scratch: Dict = {}
        self.v_render = torch.empty(1, self.H, self.W, 3, dtype=torch.float32, device=dev)
        self.v_alphas = torch.zeros(1, self.H, self.W, 1, dtype=torch.float32, device=dev)
        # (the screen-radius statistic exists only for a preset that reads it, and is written only while it is read)
        self.stats = {k: torch.zeros(cap, dtype=torch.float32, device=dev)
                      for k in ("grad2d", "count") + (("radii",) if self.cfg.refine_scale2d_stop_iter > 0 else ())}
        self.flags_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        self.count_buf = torch.empty(cap, dtype=torch.int32, device=dev)
        self.offs_buf = torch.empty(cap, dtype=torch.int32, device=dev)
        self.total_buf = torch.zeros(1, dtype=torch.int32, device=dev)
        self.gen = torch.Generator(device="cpu").manual_seed(self.cfg.seed)
        self.dev_gen = torch.Generator(device=self.device).manual_seed(self.cfg.seed + 1)   # per-step draws stay on the device
        self._bg_table: Optional[torch.Tensor] = None
        self._side_stream = None
        self._step_open = False                    # True between a step's first launch and its last: an exception leaves it set
        self._overlap_on, self._overlap_checked_at, self.refine_count = None, -1, 0
        self.last: Dict = {}
        self.last_refine: Dict = {}
        self.refine_totals: Dict = {}

    # -- helpers -------------------------------------------------------------------
    def _model_layout(self) -> Dict:
        """Keyword arguments for GaussianModel (the sharded data-parallel trainer asks for flat buffers)."""
        # One allocation per kind (parameters, exp_avg, exp_avg_sq, gradients) instead of eighteen: the fused backward +
        # Adam kernel streams all of them at once (18 separate arrays 588-589 us, flat 548-575 us on the same box,
        # profiles/r02_placement_ab.txt).  It is also what the sharded optimiser exchanges.
        return dict(flat=True)

    def _n(self) -> int:
        return self.model.n

    def _flags(self) -> int:
        f = ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC
        if self.cfg.antialiased:
            f |= ops.FLAG_ANTIALIASED
        return f

    def downscale_now(self) -> int:
        c = self.cfg
        return 2 ** max(c.num_downscales - self.step_count // max(c.resolution_schedule, 1), 0)

    def set_resolution(self, d: int) -> None:
        """Train / render at 1/d of the full resolution: (W // d, H // d), intrinsics scaled by 1/d,
        targets area-averaged over d x d blocks (nerfstudio: Cameras.rescale_output_resolution +
        resize_image's box filter [UPSTREAM-UNVERIFIED]).  Re-allocates the per-pixel buffers."""
        d = max(1, int(d))
        if d == self._cur_d:
            return
        if self._auto_cap is not None:        # intersection counts scale with the tile grid: one calibrated capacity per level
            self._auto_caps[self._cur_d], self._auto_cap_n[self._cur_d] = self._auto_cap, self.model.n
        self._cur_d = d
        self._auto_cap = None
        if d in self._auto_caps:              # back at a level seen before: its capacity, scaled by the growth in Gaussians since
            self._auto_cap = int(self._auto_caps[d] * max(1.0, self.model.n / max(self._auto_cap_n.get(d, self.model.n), 1))) + (1 << 16)
        self.W, self.H = max(1, self.W0 // d), max(1, self.H0 // d)
        self.Ks = self.Ks0.clone()
        self.Ks[:, :2, :] /= float(d)
        dev = self.device
        self.raster_out, self.loss_scratch = {}, {}
        self.v_render = torch.empty(1, self.H, self.W, 3, dtype=torch.float32, device=dev)
        self.v_alphas = torch.zeros(1, self.H, self.W, 1, dtype=torch.float32, device=dev)

    def _target(self, view_index: int) -> torch.Tensor:
        """[1,H,W,3] target of the current resolution level: float32, or uint8 at full resolution from a uint8 cache."""
        img, d = self.images[view_index], self._cur_d
        if d > 1:
            Hc, Wc = self.H * d, self.W * d                # nerfstudio's strided box filter drops the remainder
            if (Hc, Wc) != (self.H0, self.W0):
                img = img[:Hc, :Wc].contiguous()
            if img.dtype == torch.uint8:
                return ops.image_downscale_area(img, self.H, self.W, as_float=True)[None]
            return img.view(self.H, d, self.W, d, 3).mean(dim=(1, 3))[None].contiguous()
        return img[None]          # float32, or uint8 straight from the image cache (the loss kernels form value / 255 themselves)

    def sh_degree_now(self) -> int:
        return min(self.step_count // self.cfg.sh_degree_interval, self.cfg.sh_degree)

    def lrs(self):
        c = self.cfg
        t = min(self.step_count / max(c.max_steps, 1), 1.0)
        lr_means = c.lr_means * c.scene_scale * (c.lr_means_final_ratio ** t)
        return (lr_means, c.lr_quats, c.lr_scales, c.lr_opacities, c.lr_sh0, c.lr_shN)

    def _forward(self, viewmat, K, sh_degree, background=None, exact_isect=False, segments=False, hooks=None):
        m, n = self.model, self._n()
        radii, splats = self.radii[:, :n], self.splats[:, :n]
        cap = self.cfg.max_isect
        if cap is None and self.cfg.auto_isect_capacity and not exact_isect:
            cap = self._auto_cap
        # with a capacity the binning is one fused call that takes its sort keys straight from the projection
        keys = self.depth_keys[:, :n] if (self.cfg.fused_binning and cap is not None) else None
        ops.project_fwd(m.p("means"), m.p("quats"), m.p("scales"), m.p("opacities"), viewmat, K, self.W, self.H,
                        sh0=m.p("sh0"), shN=m.p("shN"), sh_degree=sh_degree, near_plane=self.cfg.near_plane,
                        far_plane=self.cfg.far_plane, flags=self._flags(), radii=radii, splats=splats, depth_keys=keys)
        if hooks and "after_project" in hooks:
            hooks["after_project"](radii, splats)
        binning = ops.bin_tiles(radii, splats, self.W, self.H, 16, max_isect=cap, tight=self.cfg.tight_tiles,
                                fused=self.cfg.fused_binning, depth_keys=keys, radii_in_records=True, want_tile_keys=False)
        if hooks and "after_binning" in hooks:
            hooks["after_binning"](radii, splats)
        # (a sample of the steps: the peak only decides when the capacity grows AHEAD of an overflow; an overflow itself sets the
        #  device error word in whatever step it happens and is absorbed at the next check.  Every step it was one more launch.)
        if self.cfg.auto_isect_capacity and self.cfg.max_isect is None and (self.step_count & 7) == 0:
            torch.maximum(self._isect_peak, binning["n_isect"], out=self._isect_peak)
        # training steps: the forward leaves checkpoints so that the backward walks long tile lists in segments
        self._seg_ws = ops.raster_seg_workspace(binning, 1, self.device, self.raster_out) if segments else None
        render, alphas, last_ids = ops.rasterize_fwd(splats, binning, self.W, self.H, 16, background, self.raster_out,
                                                     seg_ws=self._seg_ws)
        if hooks and "after_raster_fwd" in hooks:
            hooks["after_raster_fwd"](radii, splats)
        self.last_binning = binning
        return radii, splats, binning, render, alphas, last_ids

    @torch.no_grad()
    def render(self, viewmat: torch.Tensor, K: torch.Tensor, sh_degree: Optional[int] = None, background=None):
        """Render one view at full resolution: [1,H,W,3], [1,H,W,1].  (Outputs alias internal buffers.)"""
        self._settle()
        self.set_resolution(1)
        sd = self.cfg.sh_degree if sh_degree is None else sh_degree
        # (an arbitrary camera: the auto-sized capacity was measured on the training views only, so count exactly here)
        _, _, _, render, alphas, _ = self._forward(viewmat.view(1, 4, 4), K.view(1, 3, 3), sd, background, exact_isect=True)
        return render, alphas

    def _settle(self) -> None:
        """A step that raised between its launches (a Mi3dgsError, a refused fast path) leaves three invariants open that the next
        consumer relies on: v_splats clear except for what the projection backward clears itself, the segment
        workspace's counters reset by the backward that follows every forward, and the side stream's Adam ordered before the main
        stream's next reader.  Called at the head of step(), render(), refine() and by the checkpoint / export paths via
        `splats_state_dict` of the CLI; a no-op after a step that completed."""
        if not self._step_open:
            return
        torch.cuda.synchronize(self.device)       # both streams: whatever the aborted step launched has finished
        self.v_splats.zero_()
        ws = self.raster_out.get("seg_ws")
        if ws is not None:
            ops._lib.call("mi3dgs_raster_seg_workspace_init", ops._p(ws), ws.numel(), ops._stream(self.device))
        self._step_open = False

    # -- one training iteration ------------------------------------------------------
    @torch.no_grad()
    def step(self, view_index: int, want_loss: bool = False):
        c, m = self.cfg, self.model
        self._settle()
        n = self._n()
        self.set_resolution(self.downscale_now())
        if c.auto_isect_capacity and c.max_isect is None and self._auto_cap is None:
            self.calibrate_isect_capacity()
        viewmat = self.viewmats[view_index: view_index + 1]
        K = self.Ks[view_index: view_index + 1]
        gt = self._target(view_index)
        sd = self.sh_degree_now()
        bg = None
        if c.random_background:
            # one draw of 8 192 backgrounds at a time, a view of it per step (a torch.rand per step was a launch per step)
            k = self.step_count & 8191
            if self._bg_table is None or k == 0:
                self._bg_table = torch.rand(8192, 1, 3, generator=self.dev_gen, device=self.device)
            bg = self._bg_table[k]
        sreg = c.use_scale_regularization and self.step_count % c.scale_reg_every == 0
        # (a step that applies BOTH splatfacto's scale regulariser and a strategy's own regularisers takes the unfused launches)
        fused = c.fuse_adam and self._can_fuse_adam() and not (sreg and any(self._fused_regularisers()))
        # (a step that applies splatfacto's scale regulariser gives culled Gaussians a gradient too -- the only one they have:
        #  mi3dgs_adam_culled_groups forms it for the scales group itself; MCMC's regularisers touch every group and keep one launch)
        regs = self._fused_regularisers() if fused else (0.0, 0.0)
        split = (fused and c.overlap_culled_adam in ("after_project", "after_binning", "after_raster_fwd")
                 and not any(regs) and self._overlap_pays())
        hooks = None
        self._step_open = True
        if split:
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
                self._ev_main, self._ev_side = torch.cuda.Event(), torch.cuda.Event()

            def culled_groups(radii_, splats_):
                # (needs this step's radii; touches the parameters and moments of fully culled groups only, which nothing else
                #  reads or writes until the next step's projection)
                self._ev_main.record(torch.cuda.current_stream(self.device))
                self._side_stream.wait_event(self._ev_main)
                with torch.cuda.stream(self._side_stream):
                    bank_ = m.banks[m.cur]
                    ops.adam_culled_groups([bank_[g]["p"] for g in GROUPS], [bank_[g]["m"] for g in GROUPS],
                                           [bank_[g]["v"] for g in GROUPS], self.lrs(), self.step_count + 1, radii_, n=n,
                                           beta1=c.adam_beta1, beta2=c.adam_beta2, eps=c.adam_eps,
                                           scale_reg_weight=c.scale_reg_weight if sreg else 0.0, scale_reg_max_ratio=c.max_gauss_ratio)
                    self._ev_side.record(self._side_stream)

            hooks = {c.overlap_culled_adam: culled_groups}
        radii, splats, binning, render, alphas, last_ids = self._forward(viewmat, K, sd, bg, segments=c.raster_segments, hooks=hooks)
        sums, scratch = ops.loss_fwd(render, gt, self.loss_scratch, want_sums=want_loss)
        ops.loss_bwd(render, gt, scratch, c.ssim_lambda, 1.0, self.v_render)
        # (v_splats is clear: zeroed at allocation and at every refine, and the projection backward below clears the rows of the
        #  visible Gaussians -- the only rows rasterize_bwd writes -- as it reads them; the fill pass was a launch per step)
        v_splats = self.v_splats[:, :n]
        ops.rasterize_bwd(splats, binning, self.W, self.H, alphas, last_ids, self.v_render, self.v_alphas, 16, bg,
                          c.absgrad, v_splats, render=render, seg_ws=self._seg_ws)
        track = c.densify and self.step_count < c.refine_stop_iter
        stats = ({k: v[:n] for k, v in self.stats.items() if k != "radii" or self.step_count < c.refine_scale2d_stop_iter}
                 if track else None)
        bank = m.banks[m.cur]
        if fused:
            ops.project_bwd_adam([bank[g]["p"] for g in GROUPS], [bank[g]["m"] for g in GROUPS],
                                 [bank[g]["v"] for g in GROUPS], self.lrs(), self.step_count + 1, viewmat, K, self.W,
                                 self.H, radii, splats, v_splats, n=n, sh_degree=sd,
                                 flags=self._flags() | ops.FLAG_CLEAR_VSPLATS | (ops.FLAG_ONLY_VISIBLE_GROUPS if split else 0),
                                 beta1=c.adam_beta1, beta2=c.adam_beta2, eps=c.adam_eps,
                                 scale_reg_weight=c.scale_reg_weight if sreg else 0.0,
                                 scale_reg_max_ratio=c.max_gauss_ratio, stats=stats, stat_use_abs=c.absgrad,
                                 mcmc_opacity_reg=regs[0], mcmc_scale_reg=regs[1])
            if split:          # the next step's projection reads every parameter
                torch.cuda.current_stream(self.device).wait_event(self._ev_side)
        else:
            grads = {"v_" + g: m.grad(g) for g in GROUPS}
            ops.project_bwd(m.p("means"), m.p("quats"), m.p("scales"), m.p("opacities"), viewmat, K, self.W, self.H,
                            radii, splats, v_splats, sh0=m.p("sh0"), shN=m.p("shN"), color_mode=ops.COLOR_SH,
                            sh_degree=sd, flags=self._flags() | ops.FLAG_CLEAR_VSPLATS, out=grads, stats=stats, stat_use_abs=c.absgrad)
            if sreg:
                ops.scale_reg(m.p("scales"), c.scale_reg_weight, c.max_gauss_ratio, v_scales=m.grad("scales"))
            self._grad_hooks()
            self._optimizer_step(n)
        self._step_open = False
        if c.densify:
            self._strategy_post_step()
        self.last = dict(binning=binning, sums=sums)
        self.step_count += 1
        if want_loss:
            return float(ops.loss_value(sums, self.H * self.W * 3, c.ssim_lambda))
        return None

    def _overlap_pays(self) -> bool:
        """overlap_culled_adam only where whole groups ARE culled: on an object-centric scene every Gaussian is in view and the side
        launch is a launch, two events and a stream switch per step for nothing (measured: 11.1 -> 11.7 s over a 30 000-step run of
        a 40 k-Gaussian scene).  Decided from the last projection's radii where the host waits anyway (first step, every refine)."""
        if self._overlap_on is None or self._overlap_checked_at != self.refine_count:
            n = self._n()
            on = False
            if n >= self.cfg.overlap_min_gaussians and self.step_count > 0:
                vis = (self.radii[0, : n // 64 * 64] > 0).all(-1).view(-1, 64).any(1)
                on = float((~vis).float().mean()) > 0.05          # (one host sync)
            if self.step_count > 0:
                self._overlap_on, self._overlap_checked_at = on, self.refine_count
            return on
        return self._overlap_on

    def _grad_hooks(self):
        """Extra gradient terms between the backward and the optimiser (the MCMC regularisers)."""
        return

    def _optimizer_step(self, n: int):
        """Adam over all six groups from the materialised gradients.  The data-parallel trainers override this:
        gradient mean over the ranks first (parallel.py), or reduce-scatter -> Adam on a slice -> all-gather."""
        c, m = self.cfg, self.model
        bank = m.banks[m.cur]
        ops.adam_step([bank[g]["p"] for g in GROUPS], [m.grads[g] for g in GROUPS], [bank[g]["m"] for g in GROUPS],
                      [bank[g]["v"] for g in GROUPS], self.lrs(), self.step_count + 1, beta1=c.adam_beta1,
                      beta2=c.adam_beta2, eps=c.adam_eps, numel=[n * w for w in WIDTHS])

    def _can_fuse_adam(self) -> bool:
        return True

    def _fused_regularisers(self):
        """(opacity_reg, scale_reg) of a strategy whose regularisers the fused backward + Adam folds in (MCMC); (0, 0) otherwise."""
        return 0.0, 0.0

    # -- intersection capacity (auto_isect_capacity) -------------------------------------
    def calibrate_isect_capacity(self, margin: float = 2.0) -> int:
        """Bin every training view once at the current resolution (count read back: V host syncs, once) and size the
        buffers at `margin` x the worst view."""
        self._auto_cap = None
        sd = self.sh_degree_now()
        worst = 0
        for i in range(self.viewmats.shape[0]):
            _, _, binning, _, _, _ = self._forward(self.viewmats[i: i + 1], self.Ks[i: i + 1], sd)
            worst = max(worst, int(binning["n_isect"].item()))
        self._auto_cap = int(margin * worst) + (1 << 18)
        self._isect_peak.zero_()
        return self._auto_cap

    def _auto_cap_check(self, bad_bits: int = 0, scale: float = 1.0) -> None:
        """Where the host waits anyway (refine, periodic checks): grow the capacity when the running peak comes near it, or
        when a view overflowed it (its lists were truncated for the steps since the last check: reported through self.log,
        counted in self.isect_overflows, which the CLI puts into the run's stats)."""
        if not (self.cfg.auto_isect_capacity and self.cfg.max_isect is None) or self._auto_cap is None:
            return
        peak = int(self._isect_peak.item())
        self._isect_peak.zero_()
        cap = self._auto_cap
        if (bad_bits & 4) or peak >= cap:
            # the lists of some steps since the last check were cut at the capacity (their farthest splats missing): say so
            self.isect_overflows += 1
            old = cap
            cap = 2 * max(cap, peak)
            if self.log is not None:
                self.log(f"tile-list capacity overflow absorbed: steps {self._last_cap_check_step}..{self.step_count} saw up to "
                         f"{peak} intersections against a capacity of {old}; lists were truncated there, capacity now {cap} "
                         f"(overflow #{self.isect_overflows})")
        elif peak > 0.6 * cap:
            cap = int(2.0 * peak) + (1 << 18)
        self._last_cap_check_step = self.step_count
        self._auto_cap = max(cap, int(cap * scale))

    # -- DefaultStrategy.step_post_backward --------------------------------------------
    def _strategy_post_step(self):
        c, step = self.cfg, self.step_count
        if step >= c.refine_stop_iter:
            return
        if step > c.refine_start_iter and step % c.refine_every == 0 and step % c.reset_every >= c.pause_refine_after_reset:
            self.refine(do_grow=True)
        if step % c.reset_every == 0 and step > 0:
            self.reset_opacity()

    def _async_error_bits(self) -> int:
        """The device's sticky error word (synchronises).  The data-parallel trainer ORs it over the ranks."""
        return ops._lib.async_errors()

    def check_async_errors(self) -> None:
        """Raises if, since the last check, a chained kernel's bounded wait ran out (bits 1, 2: wrong
        offsets downstream) or a view produced more tile intersections than `max_isect` (bit 4: lists
        truncated).  Call it where the host waits anyway: refine, end of training, end of a benchmark."""
        bad = self._async_error_bits()
        if self.cfg.auto_isect_capacity and self.cfg.max_isect is None:
            self._auto_cap_check(bad)
            bad &= ~4
        if bad & 3:
            raise ops._lib.Mi3dgsError(f"a chained kernel gave up waiting (bits {bad:#x}); results since the last check are invalid")
        if bad & 4:
            raise ops._lib.Mi3dgsError(f"tile intersections exceeded the capacity max_isect={self.cfg.max_isect}: "
                                       f"lists were truncated since the last check (bits {bad:#x})")

    def refine(self, do_grow: bool = True) -> Dict[str, int]:
        """One densify+prune pass; returns counts.  One host sync (the new Gaussian count)."""
        c, m = self.cfg, self.model
        self._settle()
        self.check_async_errors()                    # the host waits here anyway: did every chained kernel resolve?
        self.v_splats.zero_()                        # (belt and braces: the steps keep it clear themselves, see step())
        self.refine_count += 1
        n = m.n
        st = ops._stream(self.device)
        flags, counts, offs = self.flags_buf[:n], self.count_buf[:n], self.offs_buf[:n]
        screen = "radii" in self.stats and self.step_count < c.refine_scale2d_stop_iter
        ops._lib.call("mi3dgs_densify_decide", n, ops._p(m.p("scales")), ops._p(m.p("opacities")),
                      ops._p(self.stats["grad2d"]), ops._p(self.stats["count"]), ops._p(self.stats["radii"]) if screen else None,
                      float(c.grow_grad2d), float(c.grow_scale3d * c.scene_scale), float(c.grow_scale2d), float(c.prune_opa),
                      float(c.prune_scale3d * c.scene_scale), float(c.prune_scale2d),
                      int(do_grow), int(self.step_count > c.reset_every), ops._p(flags), ops._p(counts), st)
        ops.scan_exclusive_u32(counts, offs, self.total_buf)
        new_n = int(self.total_buf.item())
        if new_n > m.capacity and do_grow:
            # out of room: this pass only prunes (upstream has no cap; ours is TrainConfig.capacity)
            return self.refine(do_grow=False)
        if new_n > m.capacity:
            raise RuntimeError(f"refine would keep {new_n} Gaussians but capacity is {m.capacity}")
        src, dst = m.banks[m.cur], m.banks[1 - m.cur]
        seed = (self.cfg.seed * 1000003 + self.step_count * 7919 + 12345) & 0xFFFFFFFF
        ops._lib.call("mi3dgs_densify_scatter", n, new_n,
                      ops._ptr_array([src[g]["p"] for g in GROUPS]), ops._ptr_array([src[g]["m"] for g in GROUPS]),
                      ops._ptr_array([src[g]["v"] for g in GROUPS]), ops._ptr_array([dst[g]["p"] for g in GROUPS]),
                      ops._ptr_array([dst[g]["m"] for g in GROUPS]), ops._ptr_array([dst[g]["v"] for g in GROUPS]),
                      ops._p(flags), ops._p(offs), m.capacity, seed, ops._p(self.count_buf), st)   # count_buf: free after the scan
        hist = torch.bincount(flags.to(torch.int64), minlength=8).tolist()            # one read-back for the log line
        # (flag 3 = duplicated AND split by the screen-size rule: three outputs)
        info = dict(n_before=n, n_after=new_n, n_dup=hist[1] + hist[3], n_split=hist[2] + hist[3],
                    n_prune=hist[4] + hist[5] + hist[6] + hist[7])
        m.cur = 1 - m.cur
        m.n = new_n
        if self._auto_cap is not None and new_n > n:
            self._auto_cap = int(self._auto_cap * (new_n / max(n, 1)))      # more Gaussians, more intersections: ahead of the peak check
        for v in self.stats.values():
            v.zero_()
        self.last_refine = info
        self.refine_totals = {k: self.refine_totals.get(k, 0) + info[k] for k in ("n_dup", "n_split", "n_prune")}
        return info

    def reset_opacity(self):
        c, m = self.cfg, self.model
        thr = 2.0 * c.prune_opa
        ops._lib.call("mi3dgs_reset_opacity", m.n, ops._p(m.p("opacities")), float(math.log(thr / (1.0 - thr))),
                      ops._p(m.state("opacities", "m")), ops._p(m.state("opacities", "v")), ops._stream(self.device))
