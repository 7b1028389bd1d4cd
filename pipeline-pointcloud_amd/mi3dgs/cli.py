"""Process-boundary drop-in for the reference's `Train-Stage1` / `Nerfstudio-Export` components.

Accepts, verbatim, the three command lines the reference builds (SURVEY.md 8b tier 1):
  ns-train <model> --timestamp train-stage-1 ... --max-num-iterations S colmap --data D
           --downscale-factor k                      source/container/src/main.py:1270-1306
  ns-export gaussian-splat --load-config outputs/unnamed/splatfacto/train-stage-1/config.yml
           --output-dir D/exports                    main.py:1455-1468
  python gsplat/examples/simple_trainer.py {default|mcmc} --max_steps S --result-dir R
           --data_factor 1 --steps_scaler x --disable_viewer --packed --batch-size 1
           --data-dir D                              main.py:1328-1338
and leaves behind the artefacts the run arm expects (main.py:2155-2161, 1392-1406):
  outputs/unnamed/splatfacto/train-stage-1/{config.yml, nerfstudio_models/}, D/exports/splat.ply,
  R/ckpts/ckpt_{step}_rank{r}.pt.
Exit code 0 on success (pipeline.py:226-232 turns anything else into a failed job).  Progress
lines carry `loss=` and `it/s`, which the log scraper whitelists, and never the words it
treats as failures (lambda/workflow_complete/workflow_complete.py:154-164,262).
Hyper-parameter defaults: SURVEY.md Appendix A [UPSTREAM-UNVERIFIED].
"""
from __future__ import annotations

import json
import math
import os
import sys
import time
from typing import Dict, List, Optional, Tuple

import torch

SPLATFACTO_MODELS = ("splatfacto", "splatfacto-big", "splatfacto-mcmc")


def say(msg: str) -> None:
    print(f"[mi3dgs] {msg}", flush=True)


# ------------------------------------------------------------------------------ argv
def parse_ns_train(argv: List[str]) -> Dict:
    """nerfstudio (tyro) style: `--a.b=c` or `--a.b c`, one dataparser sub-command."""
    if not argv:
        raise SystemExit("usage: ns-train <model> [options] colmap --data DIR [--downscale-factor K]")
    out = {"model": argv[0].lower(), "opts": {}, "dataparser": None, "data": None, "downscale": 1}
    takes_value = {"--timestamp", "--logging.local-writer.enable", "--logging.profiler", "--max-num-iterations",
                   "--pipeline.datamanager.cache-images", "--pipeline.datamanager.max-thread-workers", "--data",
                   "--downscale-factor", "--output-dir", "--experiment-name", "--pipeline.model.predict-normals",
                   "--max-gaussians", "--seed"}
    i = 1
    while i < len(argv):
        a = argv[i]
        if a in ("colmap", "nerfstudio-data", "blender-data"):
            out["dataparser"] = a
            i += 1
            continue
        if not a.startswith("--"):
            raise SystemExit(f"ns-train: unexpected argument {a!r}")
        if "=" in a:
            k, v = a.split("=", 1)
            i += 1
        elif a in takes_value or (i + 1 < len(argv) and not argv[i + 1].startswith("--") and
                                  argv[i + 1] not in ("colmap", "nerfstudio-data", "blender-data")):
            k, v = a, argv[i + 1]
            i += 2
        else:
            k, v = a, "True"
            i += 1
        out["opts"][k] = v
    o = out["opts"]
    out["data"] = o.get("--data")
    out["downscale"] = int(float(o.get("--downscale-factor", "1")))
    out["max_steps"] = int(float(o.get("--max-num-iterations", "30000")))
    out["timestamp"] = o.get("--timestamp", time.strftime("%Y-%m-%d_%H%M%S"))
    out["scale_reg"] = str(o.get("--pipeline.model.use_scale_regularization", "False")).lower() == "true"
    # nerfstudio's names for the MCMC knobs of splatfacto-mcmc (SplatfactoModelConfig)
    out["mcmc"] = {k2: float(o[k1]) for k1, k2 in (("--pipeline.model.mcmc-opacity-reg", "opacity_reg"),
                                                   ("--pipeline.model.mcmc-scale-reg", "scale_reg"),
                                                   ("--pipeline.model.noise-lr", "noise_lr")) if k1 in o}
    return out


def parse_simple_trainer(argv: List[str]) -> Dict:
    if not argv or argv[0] not in ("default", "mcmc"):
        raise SystemExit("usage: simple_trainer.py {default|mcmc} --data-dir DIR --result-dir DIR [--max_steps S] ...")
    out = {"strategy": argv[0]}
    flags = {"--disable_viewer", "--disable-viewer", "--packed", "--antialiased", "--random_bkgd", "--absgrad"}
    i = 1
    kv = {}
    while i < len(argv):
        a = argv[i]
        if a in flags:
            kv[a.lstrip("-").replace("-", "_")] = True
            i += 1
        elif "=" in a:
            k, v = a.split("=", 1)
            kv[k.lstrip("-").replace("-", "_")] = v
            i += 1
        else:
            if i + 1 >= len(argv):
                raise SystemExit(f"simple_trainer.py: option {a} needs a value")
            kv[a.lstrip("-").replace("-", "_")] = argv[i + 1]
            i += 2
    out.update(data_dir=kv.get("data_dir"), result_dir=kv.get("result_dir", "results"),
               max_steps=int(float(kv.get("max_steps", 30000))), data_factor=int(float(kv.get("data_factor", 1))),
               steps_scaler=float(kv.get("steps_scaler", 1.0)), batch_size=int(float(kv.get("batch_size", 1))),
               antialiased=bool(kv.get("antialiased", False)), random_bkgd=bool(kv.get("random_bkgd", False)),
               absgrad=bool(kv.get("absgrad", False)), max_gaussians=int(float(kv.get("max_gaussians", 8_000_000))),
               # gsplat's own names for the MCMC knobs (Config.opacity_reg / scale_reg, MCMCStrategy.noise_lr / min_opacity)
               mcmc={k2: float(kv[k1]) for k1, k2 in (("opacity_reg", "opacity_reg"), ("scale_reg", "scale_reg"),
                                                      ("strategy.noise_lr", "noise_lr"), ("strategy.min_opacity", "min_opacity"))
                     if k1 in kv})
    if not out["data_dir"]:
        raise SystemExit("simple_trainer.py: --data-dir is required")
    return out


# ---------------------------------------------------------------------------- configs
def splatfacto_config(model: str, max_steps: int, scale_reg: bool, n_train: int, capacity: int):
    from .trainer import TrainConfig
    big = model == "splatfacto-big"
    return TrainConfig(
        max_steps=max_steps, sh_degree=3, sh_degree_interval=1000, ssim_lambda=0.2,
        lr_means=1.6e-4, lr_means_final_ratio=0.01, lr_scales=5e-3, lr_quats=1e-3, lr_opacities=5e-2,
        lr_sh0=2.5e-3, lr_shN=2.5e-3 / 20, scene_scale=1.0,
        prune_opa=0.005 if big else 0.1, grow_grad2d=0.0005 if big else 0.0008, grow_scale3d=0.01, prune_scale3d=0.5,
        refine_start_iter=500, refine_stop_iter=15000, reset_every=3000, refine_every=100,
        # split_screen_size / cull_screen_size / stop_screen_size_at
        grow_scale2d=0.05, prune_scale2d=0.15, refine_scale2d_stop_iter=4000,
        pause_refine_after_reset=n_train + 100, absgrad=True, use_scale_regularization=scale_reg,
        random_background=True, capacity=capacity, auto_isect_capacity=True, spatial_sort_init=True, overlap_culled_adam="after_binning",
        num_downscales=2, resolution_schedule=3000)          # splatfacto: 1/4 -> 1/2 -> full, every 3000 steps


def simple_trainer_config(a: Dict, capacity: int):
    from .trainer import TrainConfig
    cfg = TrainConfig(max_steps=a["max_steps"], capacity=capacity, antialiased=a["antialiased"],
                      random_background=a["random_bkgd"], absgrad=a["absgrad"],
                      grow_grad2d=0.0008 if a["absgrad"] else 0.0002, scene_scale=1.1,
                      auto_isect_capacity=True, spatial_sort_init=True, overlap_culled_adam="after_binning")
    f = a["steps_scaler"]
    if f != 1.0:        # gsplat Config.adjust_steps
        import dataclasses
        sc = lambda x: max(1, int(x * f))      # noqa: E731
        cfg = dataclasses.replace(cfg, max_steps=sc(cfg.max_steps), sh_degree_interval=sc(cfg.sh_degree_interval),
                                  refine_start_iter=sc(cfg.refine_start_iter), refine_stop_iter=sc(cfg.refine_stop_iter),
                                  reset_every=sc(cfg.reset_every), refine_every=sc(cfg.refine_every))
    return cfg


# --------------------------------------------------------------------------- training
def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    mse = float(((a - b) ** 2).mean())
    return 99.0 if mse <= 1e-12 else -10.0 * math.log10(mse)


class ViewOrder:
    """Training view of global slot `s` (= step * world + rank): epoch e = s // V walks a permutation of
    the V views drawn from (seed, e) alone, so every rank derives the same order without talking, the
    ranks of one step get `world` different views, and a full epoch touches each view exactly once."""

    def __init__(self, n_views: int, seed: int):
        self.V, self.seed, self._epoch, self._perm = int(n_views), int(seed), -1, None

    def __call__(self, slot: int) -> int:
        e = slot // self.V
        if e != self._epoch:
            g = torch.Generator().manual_seed((self.seed * 1000003 + e * 7919 + 1) & 0x7FFFFFFF)
            self._perm, self._epoch = torch.randperm(self.V, generator=g).tolist(), e
        return self._perm[slot % self.V]


def run_training(data_dir: str, downscale: int, cfg, *, ctx=None, log_every: int = 100, save_steps=(),
                 on_save=None, strategy: str = "default", cap_max: int = 1_000_000, init_opacity: float = 0.1,
                 init_scale: float = 1.0, frame: str = "nerfstudio", mcmc_overrides: Optional[Dict] = None) -> Tuple[object, object, Dict]:
    """Loads the dataset, trains, evaluates.  Returns (trainer, dataset, stats)."""
    from . import dataset as ds_mod
    from . import parallel
    from .trainer import Trainer
    if not torch.cuda.is_available():
        raise SystemExit("mi3dgs: no GPU visible (set HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES); there is no CPU path")
    local = ctx.local_rank if ctx is not None else 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    t0 = time.time()
    ds = ds_mod.load_colmap_dataset(data_dir, downscale, frame=frame)
    if callable(cfg):
        cfg = cfg(ds)
    n_pts = ds.points.shape[0]
    say(f"dataset: {len(ds.train_idx)} train / {len(ds.eval_idx)} eval images {ds.width}x{ds.height}, {n_pts} SfM points")
    params = ds_mod.init_gaussians(ds.points.to(dev), ds.points_rgb, init_opacity=init_opacity, init_scale=init_scale)
    imgs = ds.load_images(ds.train_idx, dev, as_u8=True)       # device image cache (uint8)
    vm, ks = ds.viewmats[ds.train_idx].to(dev), ds.Ks[ds.train_idx].to(dev)
    if strategy == "mcmc":
        from .strategy_mcmc import MCMCConfig, MCMCTrainer
        mc = MCMCConfig(cap_max=cap_max, refine_stop_iter=max(1, int(cfg.max_steps * 25 / 30)), **(mcmc_overrides or {}))
        if ctx is not None and ctx.active:
            tr = parallel.make_mcmc_data_parallel()(params, vm, ks, imgs, ds.width, ds.height, cfg, mc, ctx=ctx)
        else:
            tr = MCMCTrainer(params, vm, ks, imgs, ds.width, ds.height, cfg, mc)
    else:
        cls = parallel.DataParallelTrainer if (ctx is not None and ctx.active) else Trainer
        kw = {"ctx": ctx} if cls is parallel.DataParallelTrainer else {}
        tr = cls(params, vm, ks, imgs, ds.width, ds.height, cfg, **kw)
    if ctx is None or ctx.rank == 0:
        tr.log = say
    say(f"loaded in {time.time() - t0:.1f}s; training {cfg.max_steps} steps from {tr.model.n} Gaussians "
        f"(capacity {tr.model.capacity})")
    V = len(ds.train_idx)
    world = ctx.world if ctx is not None and ctx.active else 1
    rank = ctx.rank if ctx is not None else 0
    view_of = ViewOrder(V, cfg.seed)
    t_train = time.time()
    t_last, s_last = t_train, 0
    from . import _lib
    prof = os.environ.get("MI3DGS_PROFILE_STEPS")          # "a:b": per-kernel table of steps [a, b) on rank 0
    prof_a, prof_b = (int(x) for x in prof.split(":")) if prof else (-1, -1)
    for step in range(cfg.max_steps):
        if rank == 0 and step == prof_a:
            _lib.profile_enable(True)
        if rank == 0 and step == prof_b:
            torch.cuda.synchronize()
            table = _lib.profile_read()
            _lib.profile_enable(False)
            tot = sum(v[1] for v in table.values())
            say(f"profile of steps {prof_a}..{prof_b}: {tot / (prof_b - prof_a):.3f} ms of kernels per step")
            for k, v in sorted(table.items(), key=lambda kv: -kv[1][1])[:14]:
                say(f"  {k:24s} {v[1] / (prof_b - prof_a):7.3f} ms/step  {v[0] / (prof_b - prof_a):5.1f} launches  {1e3 * v[1] / max(v[0], 1):8.1f} us each")
        want = (step % log_every == 0) or step == cfg.max_steps - 1
        loss = tr.step(view_of(step * world + rank), want_loss=want)
        if want:
            tr.check_async_errors()        # the host has just waited for the loss: device error word, capacity of the tile lists
        if want and rank == 0:
            now = time.time()
            rate = (step - s_last + 1) / max(now - t_last, 1e-9)
            t_last, s_last = now, step + 1
            rt = tr.refine_totals
            extra = f" (+{rt.get('n_dup', 0)} dup +{rt.get('n_split', 0)} split -{rt.get('n_prune', 0)} pruned)" if rt else ""
            say(f"step {step + 1}/{cfg.max_steps} loss={loss:.5f} gaussians={tr.model.n}{extra} | {rate:.1f} it/s")
        if (step + 1) in save_steps and on_save is not None:
            on_save(tr, ds, step)
    torch.cuda.synchronize()
    tr.check_async_errors()                # chained kernels that gave up / truncated tile lists since the last refine
    train_s = time.time() - t_train
    stats = dict(train_seconds=train_s, iters_per_sec=cfg.max_steps / max(train_s, 1e-9), gaussians=tr.model.n,
                 isect_overflows=tr.isect_overflows)
    if ds.eval_idx and rank == 0:
        ev = ds.load_images(ds.eval_idx, dev)
        ps = []
        for j, i in enumerate(ds.eval_idx):
            r, _ = tr.render(ds.viewmats[i].to(dev), ds.Ks[i].to(dev))
            ps.append(psnr(r[0].clamp(0, 1), ev[j]))
        stats["eval_psnr"] = sum(ps) / len(ps)
        say(f"eval: psnr={stats['eval_psnr']:.2f} dB over {len(ps)} held-out images")
        if os.environ.get("MI3DGS_EVAL_DETAIL"):
            say("eval per view: " + " ".join(f"{p:.1f}" for p in ps))
            trn = ds.load_images(ds.train_idx[:8], dev)
            pt = [psnr(tr.render(ds.viewmats[i].to(dev), ds.Ks[i].to(dev))[0][0].clamp(0, 1), trn[j]) for j, i in enumerate(ds.train_idx[:8])]
            say("train per view: " + " ".join(f"{p:.1f}" for p in pt))
            op = torch.sigmoid(tr.model.p("opacities")[: tr.model.n].flatten())
            sc = tr.model.p("scales")[: tr.model.n].exp().amax(-1)
            say(f"opacity quantiles {[round(float(q), 4) for q in torch.quantile(op[:1_000_000], torch.tensor([0.1, 0.5, 0.9, 0.99], device=op.device))]} "
                f"max scale quantiles {[round(float(q), 4) for q in torch.quantile(sc[:1_000_000], torch.tensor([0.5, 0.9, 0.99, 1.0], device=op.device))]}")
    if rank == 0:
        say(f"trained in {train_s:.1f}s = {stats['iters_per_sec']:.1f} it/s, {tr.model.n} Gaussians")
        if os.environ.get("MI3DGS_LIST_STATS"):          # the last step's tile lists: how long, and how far the pixels walked them
            b, ro = tr.last_binning, tr.raster_out
            I = int(b["n_isect"].item())
            offs = b["isect_offsets"].flatten().long()
            ln = torch.diff(torch.cat([offs, offs.new_tensor([I])]))
            th, tw = b["isect_offsets"].shape[-2:]
            last = ro["last_ids"][0].long()
            hit = ro["alphas"][0, ..., 0] > 0
            start = b["isect_offsets"][0].long().repeat_interleave(16, 0).repeat_interleave(16, 1)[: last.shape[0], : last.shape[1]]
            walked = torch.where(hit, last - start + 1, torch.zeros_like(last))
            wt = torch.nn.functional.max_pool2d(walked[None, None].float(), 16, ceil_mode=True)[0, 0].flatten()
            q = lambda t, f: int(torch.quantile(t.float(), f).item())      # noqa: E731
            say(f"tile lists: {I} intersections on {ln.numel()} tiles; length median {q(ln, .5)} p90 {q(ln, .9)} p99 {q(ln, .99)} max {int(ln.max())}; "
                f"walked (deepest pixel of a tile) median {q(wt, .5)} p90 {q(wt, .9)} p99 {q(wt, .99)} max {int(wt.max())}")
    return tr, ds, stats


def export_frame_splats(tr, ds) -> Dict[str, torch.Tensor]:
    """The model in the frame the upstream exporters write: the TRAINING frame, as is.  `ns-export
    gaussian-splat` dumps the model's tensors and `gsplat_pt_to_ply.py:45-73` the checkpoint's, neither maps
    back to the COLMAP world; the reference's fixed corrections (rotate_splat x:270,y:180,z:0 main.py:1481-1500,
    mirror x :1510-1523, x:180,y:180 :1556-1592) and its metric-scale stages start from that frame
    (dataset.py: frame="nerfstudio" is z-up, centred on the cameras, cameras within the unit cube)."""
    tr._settle()            # (a step that raised half-way may have left the side stream's Adam unordered against this read)
    return tr.model.splats_state_dict()


# ------------------------------------------------------------------------ entry points
def main_ns_train(argv: Optional[List[str]] = None) -> int:
    from . import _lib, io_ply
    a = parse_ns_train(sys.argv[1:] if argv is None else argv)
    if a["model"] not in SPLATFACTO_MODELS:
        raise SystemExit(f"ns-train (mi3dgs): model {a['model']!r} is not implemented; supported: {SPLATFACTO_MODELS}")
    if a["dataparser"] != "colmap" or not a["data"]:
        raise SystemExit("ns-train (mi3dgs): only `colmap --data DIR` datasets are implemented")
    mcmc = a["model"] == "splatfacto-mcmc"
    cap = int(float(a["opts"].get("--max-gaussians", 1_000_000 if mcmc else 8_000_000)))   # splatfacto-mcmc max_gs_num 1e6
    import dataclasses
    # --mi3dgs.raster-segments False: this engine's own switch (A/B of the segmented backward, docs/FINDINGS_r03.md 4.2); not an upstream option
    seg = str(a["opts"].get("--mi3dgs.raster-segments", "True")).lower() == "true"
    tr, ds, stats = run_training(a["data"], a["downscale"], lambda ds: dataclasses.replace(splatfacto_config(
        a["model"], a["max_steps"], a["scale_reg"], max(1, len(ds.train_idx)), cap), raster_segments=seg),
        strategy="mcmc" if mcmc else "default", cap_max=cap, mcmc_overrides=a.get("mcmc"))
    cfg = tr.cfg
    out_dir = os.path.join("outputs", "unnamed", "splatfacto", a["timestamp"])       # the path main.py:2158 copies from
    os.makedirs(os.path.join(out_dir, "nerfstudio_models"), exist_ok=True)
    ckpt = os.path.join(out_dir, "nerfstudio_models", f"step-{cfg.max_steps - 1:09d}.ckpt")
    io_ply.save_checkpoint(ckpt, export_frame_splats(tr, ds), cfg.max_steps - 1)
    with open(os.path.join(out_dir, "config.yml"), "w") as f:
        f.write("# mi3dgs run config (consumed by the ns-export shim)\n")
        f.write(json.dumps(dict(engine="mi3dgs", model=a["model"], data=os.path.abspath(a["data"]),
                                checkpoint=os.path.abspath(ckpt), max_steps=cfg.max_steps, stats=stats), indent=1) + "\n")
    if a["model"] != "splatfacto":
        alt = os.path.join("outputs", "unnamed", a["model"])
        os.makedirs(alt, exist_ok=True)
        link = os.path.join(alt, a["timestamp"])
        if not os.path.lexists(link):
            os.symlink(os.path.abspath(out_dir), link)
    say(f"wrote {out_dir}/config.yml and {ckpt}")
    return 0


def main_ns_export(argv: Optional[List[str]] = None) -> int:
    from . import io_ply
    argv = sys.argv[1:] if argv is None else argv
    if not argv or argv[0] != "gaussian-splat":
        raise SystemExit("ns-export (mi3dgs): only `gaussian-splat` is implemented")
    kv, i = {}, 1
    while i < len(argv):
        if "=" in argv[i]:
            k, v = argv[i].split("=", 1); i += 1
        else:
            k, v = argv[i], argv[i + 1]; i += 2
        kv[k.replace("_", "-")] = v
    cfg_path, out_dir = kv.get("--load-config"), kv.get("--output-dir")
    if not cfg_path or not out_dir:
        raise SystemExit("usage: ns-export gaussian-splat --load-config CONFIG.yml --output-dir DIR")
    txt = "".join(l for l in open(cfg_path) if not l.startswith("#"))
    cfg = json.loads(txt)
    ck = io_ply.load_checkpoint(cfg["checkpoint"])
    n = io_ply.write_ply(os.path.join(out_dir, "splat.ply"), ck["splats"])
    say(f"wrote {os.path.join(out_dir, 'splat.ply')} ({n} Gaussians)")
    return 0


def _batch_scaled(cfg, batch: int):
    """The lr / eps / beta part of gsplat's batch-size rule; the step counts are NOT divided again: the
    reference already passes --steps_scaler 1/G (main.py:1323)."""
    import dataclasses
    from . import parallel
    b = parallel.batch_scaled_config(cfg, batch)
    return dataclasses.replace(cfg, lr_means=b.lr_means, lr_scales=b.lr_scales, lr_quats=b.lr_quats,
                               lr_opacities=b.lr_opacities, lr_sh0=b.lr_sh0, lr_shN=b.lr_shN, adam_eps=b.adam_eps,
                               adam_beta1=b.adam_beta1, adam_beta2=b.adam_beta2)


def _simple_trainer_rank(rank: int, world: int, port: int, a: Dict) -> None:
    from . import io_ply, parallel
    if world > 1:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
    ctx = parallel.init_from_env() if world > 1 else parallel.DistContext(0, 1, 0)
    cfg0 = simple_trainer_config(a, a["max_gaussians"])
    bs = world * a["batch_size"]

    def cfg(ds):
        import dataclasses
        c = dataclasses.replace(cfg0, scene_scale=1.1 * ds.scene_scale)      # gsplat: parser.scene_scale * 1.1 * global_scale
        return _batch_scaled(c, bs) if bs > 1 else c

    ck_dir = os.path.join(a["result_dir"], "ckpts")

    def save(tr, ds, step):
        io_ply.save_checkpoint(os.path.join(ck_dir, f"ckpt_{step}_rank{rank}.pt"), export_frame_splats(tr, ds), step)

    save_steps = {max(1, int(s * a["steps_scaler"])) for s in (7000, 30000)}
    tr, ds, stats = run_training(a["data_dir"], a["data_factor"], cfg, ctx=ctx if world > 1 else None, frame="gsplat",
                                 save_steps=save_steps, on_save=save, strategy=a["strategy"],
                                 cap_max=min(a["max_gaussians"], 1_000_000) if a["strategy"] == "mcmc" else a["max_gaussians"],
                                 # gsplat's `mcmc` preset: init_opa 0.5, init_scale 0.1 [UPSTREAM-UNVERIFIED]
                                 init_opacity=0.5 if a["strategy"] == "mcmc" else 0.1,
                                 init_scale=0.1 if a["strategy"] == "mcmc" else 1.0, mcmc_overrides=a.get("mcmc"))
    cfg = tr.cfg
    save(tr, ds, cfg.max_steps - 1)            # every rank holds the full (replicated) model
    # The reference's exporter converts sorted(os.listdir(ckpts))[-1] (gsplat_pt_to_ply.py:36-40), a
    # LEXICOGRAPHIC sort: with scaled step counts an intermediate "ckpt_874_rank0.pt" sorts after the
    # final "ckpt_3749_rank0.pt".  Drop this rank's intermediates that would shadow the final one.
    final_name = f"ckpt_{cfg.max_steps - 1}_rank{rank}.pt"
    for s_ in save_steps:
        name = f"ckpt_{s_ - 1}_rank{rank}.pt"
        if name != final_name and name > f"ckpt_{cfg.max_steps - 1}_rank" and os.path.isfile(os.path.join(ck_dir, name)):
            os.remove(os.path.join(ck_dir, name))
    if rank == 0:
        os.makedirs(os.path.join(a["result_dir"], "stats"), exist_ok=True)
        with open(os.path.join(a["result_dir"], "stats", f"val_step{cfg.max_steps - 1:04d}.json"), "w") as f:
            json.dump(stats, f)
        say(f"wrote {ck_dir}/ckpt_{cfg.max_steps - 1}_rank*.pt")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def _visible_gpus() -> int:
    """GPUs this launcher spreads over (MI3DGS_FAKE_DEVICE_COUNT: launcher tests on a host without GPUs)."""
    fake = os.environ.get("MI3DGS_FAKE_DEVICE_COUNT")
    return int(fake) if fake else torch.cuda.device_count()


def main_simple_trainer(argv: Optional[List[str]] = None, rank_fn=None) -> int:
    """Like gsplat's own launcher: one process per visible GPU (it ignores torchrun's env).  rank_fn (tests): what
    each spawned process runs instead of _simple_trainer_rank, same signature (rank, world, port, args)."""
    rank_fn = rank_fn or _simple_trainer_rank
    a = parse_simple_trainer(sys.argv[1:] if argv is None else argv)
    world = _visible_gpus()
    if os.environ.get("MI3DGS_SINGLE_GPU") and world > 1:
        # Forced single-GPU run on a multi-GPU box.  The reference still passes --steps_scaler 1/G
        # (main.py:1322-1327: G images per step upstream, so 1/G of the steps); with one image per step
        # that would train 1/G of the job, so the division is undone here.
        a["steps_scaler"] = min(1.0, a["steps_scaler"] * world)
        say(f"training on one of the {world} visible GPUs; --steps_scaler reset to {a['steps_scaler']:g} (one image per step)")
        world = 1
    if world <= 1:
        rank_fn(0, 1, 0, a)
        return 0
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(rank_fn, args=(world, port, a), nprocs=world, join=True)
    return 0
