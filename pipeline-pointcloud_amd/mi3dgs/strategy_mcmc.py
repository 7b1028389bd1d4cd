"""MCMC densification strategy (gsplat MCMCStrategy; "3D Gaussian Splatting as Markov Chain
Monte Carlo"), what the reference selects with MODEL=splatfacto-mcmc
(source/container/src/main.py:1285-1291) or `simple_trainer.py mcmc` (main.py:1324-1327).

Per step, after Adam: positional noise on nearly transparent Gaussians.  Every `refine_every`
steps between refine_start_iter and refine_stop_iter: dead Gaussians (opacity <= min_opacity)
are teleported onto live ones sampled in proportion to opacity, then the set grows by 5 % up to
`cap_max`; sources and copies get the relocation opacity / scale, sources lose their Adam
state.  Loss adds opacity_reg * mean(opacity) + scale_reg * mean(scale).
Defaults [UPSTREAM-UNVERIFIED, SURVEY.md Appendix A].  The two arithmetic kernels are HIP
(csrc/densify.hip); multinomial sampling and row copies are tensor plumbing, every 100 steps.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Optional

import torch

from . import ops
from .trainer import GROUPS, TrainConfig, Trainer

N_MAX = 51
_NAN_CHECK = bool(__import__("os").environ.get("MI3DGS_NAN_CHECK"))
_REFINE_LOG = bool(__import__("os").environ.get("MI3DGS_MCMC_LOG"))      # one line per refine: dead fraction, opacity, ratios


@dataclasses.dataclass
class MCMCConfig:
    cap_max: int = 1_000_000
    noise_lr: float = 5e5
    min_opacity: float = 0.005
    opacity_reg: float = 0.01
    scale_reg: float = 0.01
    refine_start_iter: int = 500
    refine_stop_iter: int = 25_000
    refine_every: int = 100


def binom_table(device) -> torch.Tensor:
    t = torch.zeros(N_MAX, N_MAX, dtype=torch.float32)
    for n in range(N_MAX):
        for k in range(n + 1):
            t[n, k] = math.comb(n, k)
    return t.to(device)


def compute_relocation(opacities: torch.Tensor, scales: torch.Tensor, ratios: torch.Tensor, binoms: torch.Tensor):
    """ACTIVATED opacities[n], scales[n,3], ratios[n] int32 -> (new opacities, new scales)."""
    n = opacities.shape[0]
    no, ns = torch.empty_like(opacities), torch.empty_like(scales)
    ops._lib.call("mi3dgs_mcmc_relocation", n, ops._p(opacities.contiguous()), ops._p(scales.contiguous()),
                  ops._p(ratios.to(torch.int32).contiguous()), ops._p(binoms), ops._p(no), ops._p(ns),
                  ops._stream(opacities.device))
    return no, ns


class MCMCTrainer(Trainer):
    def __init__(self, params, viewmats, Ks, images, width, height, cfg: Optional[TrainConfig] = None,
                 mcmc: Optional[MCMCConfig] = None):
        # (the regularisers are folded into the fused backward + Adam: ops.project_bwd_adam(mcmc_*); fuse_adam=False in the config
        #  keeps the three-launch path -- backward, mi3dgs_mcmc_regularise, Adam -- which the data-parallel trainer needs anyway)
        cfg = dataclasses.replace(cfg or TrainConfig(), densify=False)
        self.mcmc = mcmc or MCMCConfig()
        if cfg.capacity is None or cfg.capacity < self.mcmc.cap_max:
            cfg = dataclasses.replace(cfg, capacity=max(self.mcmc.cap_max, params["means"].shape[0]))
        super().__init__(params, viewmats, Ks, images, width, height, cfg)
        self.binoms = binom_table(self.device)
        self.tgen = torch.Generator(device=self.device).manual_seed(cfg.seed + 77)
        self.mcmc_totals = dict(relocated=0, added=0)
        self.last_max_ratio = 0

    def _fused_regularisers(self):
        return float(self.mcmc.opacity_reg), float(self.mcmc.scale_reg)

    # gradients of the two regularisers, between the backward and Adam (unfused path)
    def _grad_hooks(self):
        m, c = self.model, self.mcmc
        ops._lib.call("mi3dgs_mcmc_regularise", m.n, ops._p(m.p("opacities")), ops._p(m.p("scales")), float(c.opacity_reg),
                      float(c.scale_reg), ops._p(m.grad("opacities")), ops._p(m.grad("scales")), ops._stream(self.device))

    @torch.no_grad()
    def step(self, view_index: int, want_loss: bool = False):
        step = self.step_count
        if _NAN_CHECK and not getattr(self, "_nan_seen", False):
            self._prev = {g: self._rows(g)[: self.model.n].clone() for g in GROUPS}
            self._prev_mv = {g: (self.model.state(g, "m")[: self.model.n].clone(), self.model.state(g, "v")[: self.model.n].clone()) for g in ("means", "scales")}
        out = super().step(view_index, want_loss)
        if _NAN_CHECK and not getattr(self, "_nan_seen", False):
            bad = ~torch.isfinite(self._rows("means")[: self.model.n]).all(1)
            if bool(bad.any()):
                i = int(torch.nonzero(bad).flatten()[0])
                print(f"[nan-check] step {step} view {view_index} BEFORE noise/refine: row {i}: prev means {self._prev['means'][i].tolist()} quats {self._prev['quats'][i].tolist()} "
                      f"log-scales {self._prev['scales'][i].tolist()} opac {self._prev['opacities'][i].tolist()}\n   grads: means {self.model.grad('means')[i].tolist()} quats {self.model.grad('quats')[i].tolist()} "
                      f"scales {self.model.grad('scales')[i].tolist()} opac {self.model.grad('opacities')[i].tolist()}\n   radii {self.radii[0, i].tolist()} record {self.splats[0, i].tolist()}\n   vrecord {self.v_splats[0, i].tolist()}"
                      f"\n   prev m/v means {self._prev_mv['means'][0][i].tolist()} {self._prev_mv['means'][1][i].tolist()} viewmat {self.viewmats[view_index].tolist()}", flush=True)
        c, m = self.mcmc, self.model
        if c.refine_start_iter < step < c.refine_stop_iter and step % c.refine_every == 0:
            n_rel = self.relocate()
            n_add = self.add_new()
            self.mcmc_totals["relocated"] += n_rel
            self.mcmc_totals["added"] += n_add
            if _REFINE_LOG and (step % (10 * c.refine_every) == 0 or n_rel > 0.2 * m.n):
                op = torch.sigmoid(self._rows("opacities")[: m.n, 0])
                q = torch.quantile(op[:1_000_000], torch.tensor([0.1, 0.5, 0.9], device=op.device)).tolist()
                sc = self._rows("scales")[: m.n].exp().amax(-1)
                print(f"[mcmc] step {step} n={m.n} relocated={n_rel} ({100.0 * n_rel / max(m.n, 1):.1f} %) added={n_add} "
                      f"max_ratio={self.last_max_ratio} opacity q10/50/90 {q[0]:.4f}/{q[1]:.4f}/{q[2]:.4f} mean {float(op.mean()):.4f} "
                      f"max-scale median {float(sc.median()):.5f} max {float(sc.max()):.4f}", flush=True)
        seed = (self.cfg.seed * 2654435761 + step * 40503 + 17) & 0xFFFFFFFF
        lr_means = self.lrs()[0]
        ops._lib.call("mi3dgs_mcmc_inject_noise", m.n, ops._p(m.p("means")), ops._p(m.p("quats")), ops._p(m.p("scales")),
                      ops._p(m.p("opacities")), float(lr_means * c.noise_lr), seed, ops._stream(self.device))
        self.refine_totals = dict(n_dup=self.mcmc_totals["added"], n_split=self.mcmc_totals["relocated"], n_prune=0)
        if _NAN_CHECK and not getattr(self, "_nan_seen", False):
            for g in GROUPS:
                bad = ~torch.isfinite(self._rows(g)[: m.n]).reshape(m.n, -1).all(1)
                if bool(bad.any()):
                    idx = torch.nonzero(bad).flatten()[:5].tolist()
                    print(f"[nan-check] step {step}: {int(bad.sum())} non-finite rows in {g}, first {idx}; refine step: {step % c.refine_every == 0}; "
                          f"opacity logits {self._rows('opacities')[idx, 0].tolist()} scales {self._rows('scales')[idx].tolist()}", flush=True)
                    self._nan_seen = True
        return out

    # -- helpers -------------------------------------------------------------------
    def _rows(self, g: str, k: str = "p") -> torch.Tensor:
        return self.model.banks[self.model.cur][g][k]

    def _apply_relocation(self, sampled: torch.Tensor, zero_source_state: bool = True):
        """New opacity / scale for the sampled sources (ratio = times sampled + 1).  gsplat's `relocate` also zeroes
        the sources' Adam state, its `sample_add` does NOT (only the appended copies start from zero moments): a
        source whose moments are zeroed takes a ~3 x lr step in every coordinate at the next Adam update (the bias
        correction is the global step's), and growth samples 5 % of the most opaque Gaussians every 100 steps.
        [UPSTREAM-UNVERIFIED]"""
        m, c = self.model, self.mcmc
        op = torch.sigmoid(self._rows("opacities")[sampled, 0])
        sc = torch.exp(self._rows("scales")[sampled])
        counts = torch.bincount(sampled, minlength=m.n)[sampled] + 1
        if _REFINE_LOG and zero_source_state:
            self.last_max_ratio = int(counts.max())               # of the relocation (the growth step's ratios are 1-2)
        no, ns = compute_relocation(op, sc, counts.clamp(max=N_MAX), self.binoms)
        # (the alternating binomial sum of the scale formula cancels in float32 for large ratios: keep the old scale rather than
        #  the logarithm of a non-positive number)
        ns = torch.where(torch.isfinite(ns) & (ns > 0), ns, sc)
        no = torch.where(torch.isfinite(no), no, op)
        no = no.clamp(min=c.min_opacity, max=1.0 - 1e-7)
        self._rows("opacities")[sampled, 0] = torch.log(no / (1.0 - no))
        self._rows("scales")[sampled] = torch.log(ns)
        if zero_source_state:
            for g in GROUPS:
                self._rows(g, "m")[sampled] = 0.0
                self._rows(g, "v")[sampled] = 0.0

    def _sync_optimizer_state(self):
        """Hook: the sharded data-parallel trainer brings every rank's Adam moments up to date here."""
        return

    def relocate(self) -> int:
        m, c = self.model, self.mcmc
        n = m.n
        self._sync_optimizer_state()
        op = torch.sigmoid(self._rows("opacities")[:n, 0])
        dead = op <= c.min_opacity
        n_dead = int(dead.sum())
        if n_dead == 0 or n_dead == n:
            return 0
        alive_idx = torch.nonzero(~dead).flatten()
        pick = torch.multinomial(op[alive_idx], n_dead, replacement=True, generator=self.tgen)
        sampled = alive_idx[pick]
        self._apply_relocation(sampled)
        dead_idx = torch.nonzero(dead).flatten()
        for g in GROUPS:
            self._rows(g)[dead_idx] = self._rows(g)[sampled]
        return n_dead

    def add_new(self) -> int:
        m, c = self.model, self.mcmc
        n = m.n
        target = min(c.cap_max, m.capacity, int(1.05 * n))
        n_new = max(0, target - n)
        if n_new == 0:
            return 0
        op = torch.sigmoid(self._rows("opacities")[:n, 0])
        self._sync_optimizer_state()
        sampled = torch.multinomial(op, n_new, replacement=True, generator=self.tgen)
        self._apply_relocation(sampled, zero_source_state=False)
        for g in GROUPS:
            self._rows(g)[n:n + n_new] = self._rows(g)[sampled]
            self._rows(g, "m")[n:n + n_new] = 0.0
            self._rows(g, "v")[n:n + n_new] = 0.0
        m.n = n + n_new
        return n_new
