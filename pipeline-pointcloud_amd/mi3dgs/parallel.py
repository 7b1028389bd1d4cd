"""Multi-GPU modes of the training path, one process per GPU over RCCL.

The reference's multi-GPU branch (source/container/src/main.py:1318-1347) runs gsplat's
`simple_trainer.py --steps_scaler 1/G --packed --batch-size 1` with `WORLD_SIZE`, `RANK`,
`LOCAL_RANK`, `MASTER_ADDR`, `MASTER_PORT` exported at main.py:623-655.  Two modes here:

* scene-per-GPU (BASELINE.json configs[3]): independent replicas, no data-path collective;
  only a barrier brackets the timed region (bench.py).
* one big scene (configs[4]): Gaussians REPLICATED on every GPU (6 M x 944 B = 5.7 GB, nothing
  against 288 GB), each rank renders a different image per step, then
      gradients      : all-reduce, mean over ranks   (every step)
      densify stats  : all-reduce sum / sum / max    (only when a refine pass runs)
  so every replica applies the identical Adam step and takes identical densify decisions
  (same counter-based RNG seed for split samples).  Upstream instead shards Gaussians and
  all-to-alls projected splats; replication trades 2 * 7/8 * |grad| of xGMI traffic per
  step for zero forward/backward exchange.

`backend="nccl"` is RCCL on ROCm; the CPU tests run the same code over `gloo`.
"""
from __future__ import annotations

import dataclasses
import math
import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import ops
from .trainer import GROUPS, WIDTHS, TrainConfig, Trainer


@dataclasses.dataclass
class DistContext:
    rank: int
    world: int
    local_rank: int
    group: Optional[object] = None

    @property
    def active(self) -> bool:
        return self.world > 1


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None) -> DistContext:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (what the reference
    exports at main.py:623-655).  Single-process runs get an inactive context."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")       # reference default, main.py:640
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return DistContext(rank, world, local)


def views_for_step(step: int, n_views: int, ctx: DistContext) -> int:
    """View this rank trains on at global step `step`: a batch of `world` consecutive views of a
    fixed round-robin order, one per rank, so a full pass touches every view exactly once."""
    return (step * ctx.world + ctx.rank) % n_views


def allreduce_mean_(tensors: Sequence[torch.Tensor], ctx: DistContext) -> None:
    """In-place mean over ranks.  One collective per tensor, issued back to back (RCCL fuses
    them on the stream); tensors must be contiguous."""
    if not ctx.active:
        return
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=ctx.group)
    inv = 1.0 / ctx.world
    for t in tensors:
        t.mul_(inv)


def allreduce_stats_(stats: Dict[str, torch.Tensor], ctx: DistContext) -> None:
    """Densify statistics: gradient-norm accumulators and visibility counts add up, screen
    radii take the max (gsplat DefaultStrategy state)."""
    if not ctx.active:
        return
    dist.all_reduce(stats["grad2d"], op=dist.ReduceOp.SUM, group=ctx.group)
    dist.all_reduce(stats["count"], op=dist.ReduceOp.SUM, group=ctx.group)
    if "radii" in stats:
        dist.all_reduce(stats["radii"], op=dist.ReduceOp.MAX, group=ctx.group)


def batch_scaled_config(cfg: TrainConfig, batch: int) -> TrainConfig:
    """gsplat simple_trainer's batch-size scaling rule [UPSTREAM-UNVERIFIED] for BS = batch *
    world: lr *= sqrt(BS), eps /= sqrt(BS), beta_i -> 1 - BS (1 - beta_i), and every step
    count is divided by BS (the reference passes --steps_scaler 1/G, main.py:1323)."""
    if batch <= 1:
        return cfg
    s = math.sqrt(batch)
    steps = lambda x: max(1, int(round(x / batch)))      # noqa: E731
    return dataclasses.replace(
        cfg,
        lr_means=cfg.lr_means * s, lr_scales=cfg.lr_scales * s, lr_quats=cfg.lr_quats * s,
        lr_opacities=cfg.lr_opacities * s, lr_sh0=cfg.lr_sh0 * s, lr_shN=cfg.lr_shN * s,
        adam_eps=cfg.adam_eps / s,
        adam_beta1=max(0.0, 1.0 - batch * (1.0 - cfg.adam_beta1)),
        adam_beta2=max(0.0, 1.0 - batch * (1.0 - cfg.adam_beta2)),
        max_steps=steps(cfg.max_steps), sh_degree_interval=steps(cfg.sh_degree_interval),
        refine_start_iter=steps(cfg.refine_start_iter), refine_stop_iter=steps(cfg.refine_stop_iter),
        reset_every=steps(cfg.reset_every), refine_every=steps(cfg.refine_every))


def _backend(group=None) -> str:
    return dist.get_backend(group)


def reduce_scatter_sum_(flat: torch.Tensor, ctx: DistContext) -> torch.Tensor:
    """Sum `flat` (numel divisible by world) over the ranks; returns this rank's 1/world slice (a view of `flat`,
    in place).  RCCL: one reduce_scatter; gloo (CPU rehearsal, tests) has none: all_reduce and take the slice."""
    L = flat.numel() // ctx.world
    mine = flat[ctx.rank * L: (ctx.rank + 1) * L]
    if _backend(ctx.group) == "nccl":
        dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM, group=ctx.group)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
    return mine


def all_gather_slices_(flat: torch.Tensor, ctx: DistContext) -> None:
    """Every rank contributes its 1/world slice of `flat`; afterwards all ranks hold the whole buffer (in place)."""
    L = flat.numel() // ctx.world
    mine = flat[ctx.rank * L: (ctx.rank + 1) * L]
    if _backend(ctx.group) == "nccl":
        dist.all_gather_into_tensor(flat, mine, group=ctx.group)
    else:
        parts = [flat[r * L: (r + 1) * L] for r in range(ctx.world)]
        dist.all_gather(parts, mine.clone(), group=ctx.group)


class DataParallelMixin:
    """Replicated-Gaussian data parallelism over a Trainer (default or MCMC strategy): identical parameters on
    every rank, one view per rank per step.

    shard_optimizer=True: gradients, parameters and moments live in flat buffers;
        per step   reduce_scatter(gradients, sum)  ->  Adam on this rank's 1/G slice  ->  all_gather(parameters)
        at refine  all_gather(exp_avg), all_gather(exp_avg_sq) (each rank only keeps its slice current), all-reduce
                   of the densify statistics (sum, sum, max); then the identical surgery everywhere.
      xGMI bytes per rank and step: 2 x (G-1)/G x 236 B x live rows  (6 M Gaussians, G = 8: 2.5 GB), against
      2 x that for the dense all-reduce, and 1/G of the Adam traffic (1 652 B per Gaussian) instead of all of it.
    shard_optimizer=False (round 1): one dense all-reduce per group, the full Adam step on every rank."""

    def _dp_init(self, ctx: Optional[DistContext], shard_optimizer: bool):
        self.ctx = ctx or init_from_env()
        self.shard_optimizer = bool(shard_optimizer) and self.ctx.active
        self._moments_current = True            # False once a sharded Adam step has run since the last sync

    def _model_layout(self) -> Dict:
        ctx = getattr(self, "_pending_ctx", None)
        world = ctx.world if ctx is not None else 1
        if getattr(self, "_pending_shard", False) and ctx is not None and ctx.active:
            return dict(flat=True, align=4 * world)            # slices stay 16-byte aligned
        return {}

    def _live_grads(self) -> List[torch.Tensor]:
        m = self.model
        return [m.grads[g][: m.n] for g in GROUPS]

    def _can_fuse_adam(self) -> bool:
        return not self.ctx.active          # the gradients have to exist to be reduced

    # -- the optimiser step --------------------------------------------------------------
    # The exchange is per parameter group over the LIVE rows: rank r owns rows [r R / G, (r + 1) R / G) of every
    # group, R = the live count rounded up to 4 G rows (so that every slice of every width starts 16-byte aligned; the model's
    # capacity is a multiple of 4 G).  Round 2 cut the flat buffers by CAPACITY: with n < capacity (the normal case: capacity =
    # max_gaussians) the high ranks owned nothing but padding, a few ranks did all of the Adam work (shN is 45/59 of the
    # buffer), and the padding travelled over xGMI every step.
    def _exchange_rows(self) -> int:
        m, q = self.model, 4 * self.ctx.world
        return min(m.capacity, (m.n + q - 1) // q * q)

    def _group_spans(self):
        """(group index, first float of the group in the flat buffers, floats of its exchanged rows)."""
        m, R = self.model, self._exchange_rows()
        out, off = [], 0
        for gi, w in enumerate(WIDTHS):
            out.append((gi, off, w * R))
            off += w * m.capacity
        return out

    def _slice_pieces(self):
        """This rank's part of the flat buffers as (group index, first float, count) pieces: its rows of every group, cut at
        the live count (rows beyond n are alignment padding: exchanged, never updated)."""
        m, G, r = self.model, self.ctx.world, self.ctx.rank
        L = self._exchange_rows() // G
        lo, hi = r * L, min((r + 1) * L, m.n)
        if hi <= lo:
            return []
        return [(gi, off + w * lo, w * (hi - lo)) for (gi, off, _), w in zip(self._group_spans(), WIDTHS)]

    def _optimizer_step(self, n: int):
        if not self.ctx.active:
            return super()._optimizer_step(n)
        c, m = self.cfg, self.model
        if not self.shard_optimizer:
            allreduce_mean_(self._live_grads(), self.ctx)
            return super()._optimizer_step(n)
        gflat = m.flat["g"]
        for _, off, cnt in self._group_spans():
            reduce_scatter_sum_(gflat[off: off + cnt], self.ctx)
        pieces = self._slice_pieces()
        if pieces:
            lrs = self.lrs()
            P, M, V = m.flat["p"][m.cur], m.flat["m"][m.cur], m.flat["v"][m.cur]
            # the reduce-scatter delivered the SUM over the ranks; the mean's 1 / world is applied by the Adam kernel as it
            # reads the gradient (a separate mul_ over the slice was one more read and write of it per step)
            ops.adam_step([P[a: a + cnt] for _, a, cnt in pieces], [gflat[a: a + cnt] for _, a, cnt in pieces],
                          [M[a: a + cnt] for _, a, cnt in pieces], [V[a: a + cnt] for _, a, cnt in pieces],
                          [lrs[gi] for gi, _, _ in pieces], self.step_count + 1,
                          beta1=c.adam_beta1, beta2=c.adam_beta2, eps=c.adam_eps, grad_scale=1.0 / self.ctx.world)
        P = m.flat["p"][m.cur]
        for _, off, cnt in self._group_spans():
            all_gather_slices_(P[off: off + cnt], self.ctx)
        self._moments_current = False

    def _sync_optimizer_state(self):
        """Bring exp_avg / exp_avg_sq up to date on every rank (each keeps only its rows current between refines)."""
        if self.ctx.active and self.shard_optimizer and not self._moments_current:
            m = self.model
            for k in ("m", "v"):
                buf = m.flat[k][m.cur]
                for _, off, cnt in self._group_spans():
                    all_gather_slices_(buf[off: off + cnt], self.ctx)
            self._moments_current = True

    def xgmi_bytes_per_step(self) -> int:
        """Bytes this rank sends (= receives) per training step for the gradient / parameter exchange."""
        if not self.ctx.active:
            return 0
        G, m = self.ctx.world, self.model
        if self.shard_optimizer:
            return int(2 * (G - 1) / G * 4 * sum(WIDTHS) * self._exchange_rows())
        return int(2 * 2 * (G - 1) / G * 4 * sum(WIDTHS) * m.n)           # ring all-reduce = reduce-scatter + all-gather of everything

    def _async_error_bits(self) -> int:
        # every rank must take the same decision, or the ones that do not raise hang in the next collective
        bad = super()._async_error_bits()
        if self.ctx.active:
            dev = self.device if _backend(self.ctx.group) == "nccl" else "cpu"
            t = torch.tensor([(bad >> b) & 1 for b in range(8)], dtype=torch.int32, device=dev)   # RCCL has no bitwise OR
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.ctx.group)
            bad = sum(int(v) << b for b, v in enumerate(t.tolist()))
        return bad

    def refine(self, do_grow: bool = True):
        n = self.model.n
        self._sync_optimizer_state()
        allreduce_stats_({k: v[:n] for k, v in self.stats.items()}, self.ctx)
        return super().refine(do_grow)

    def step_global(self, global_step: Optional[int] = None, want_loss: bool = False):
        s = self.step_count if global_step is None else global_step
        return self.step(views_for_step(s, self.viewmats.shape[0], self.ctx), want_loss)

    def replicas_in_sync(self) -> bool:
        """True when every rank holds bit-identical parameters (debug / tests)."""
        if not self.ctx.active:
            return True
        ok = True
        for g in GROUPS:
            p = self.model.p(g).detach().reshape(-1)
            lo, hi = p.clone(), p.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.ctx.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.ctx.group)
            ok = ok and bool(torch.equal(lo, hi))
        return ok


class DataParallelTrainer(DataParallelMixin, Trainer):
    def __init__(self, *args, ctx: Optional[DistContext] = None, shard_optimizer: bool = True, **kw):
        self._pending_ctx, self._pending_shard = ctx or init_from_env(), shard_optimizer
        super().__init__(*args, **kw)
        self._dp_init(self._pending_ctx, shard_optimizer)


def make_mcmc_data_parallel():
    """MCMC strategy over the same data-parallel machinery (class built lazily: strategy_mcmc imports trainer)."""
    from .strategy_mcmc import MCMCTrainer

    class MCMCDataParallelTrainer(DataParallelMixin, MCMCTrainer):
        def __init__(self, *args, ctx: Optional[DistContext] = None, shard_optimizer: bool = True, **kw):
            self._pending_ctx, self._pending_shard = ctx or init_from_env(), shard_optimizer
            super().__init__(*args, **kw)
            self._dp_init(self._pending_ctx, shard_optimizer)

    return MCMCDataParallelTrainer
