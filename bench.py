#!/usr/bin/env python3
"""Headline benchmark: 3DGS training iterations/s (+ render FPS) at 1080p on MI355X.

Contract (one JSON line on stdout from rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver with torch.distributed.run, one rank per GPU; every rank
  trains its own independent scene (BASELINE.json configs[3]: "8 independent scenes, one per
  GPU") -> weak scaling, no data-path collective, RCCL barrier + max-over-ranks timing.

A "step" is one full training iteration on one 1920x1080 view of the S2 "garden-like"
synthetic scene (2 M Gaussians, SH degree 3): project+SH -> tile binning ->
rasterise -> L1+SSIM loss -> rasterise backward -> project backward (+ densify statistics)
-> Adam.  Inputs (Gaussians, cameras, target images) are resident in HBM before timing.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
F32_PEAK_TFLOPS = 157.3      # f32 MFMA == f32 vector peak


# the reference call site each preset stands for
PRESET_JOBS = {
    "splatfacto": "ns-train splatfacto --pipeline.model.use_scale_regularization=True (source/container/src/main.py:1270-1306): absgrad, "
                  "scale regulariser every 10 steps, random background, screen-size statistics, prune_opa 0.1 / grow_grad2d 8e-4",
    "simple_trainer": "gsplat examples/simple_trainer.py default (source/container/src/main.py:1318-1347): DefaultStrategy defaults, no absgrad, "
                      "no regulariser, no background",
}

_T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="garden", choices=["cube", "lego", "garden", "6m"])
    ap.add_argument("--gaussians", dest="n", type=int, default=None, help="override the Gaussian count")
    ap.add_argument("--views", type=int, default=8, help="target views kept resident")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "scene-shard"],
                    help="replicas (default, BASELINE configs[3]): one independent scene per GPU, weak scaling.  scene-shard "
                         "(configs[4]): ONE scene, Gaussians replicated, one view per rank per step, gradients reduce-scattered, "
                         "Adam on a 1/G slice, parameters all-gathered; value = training views per second")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-profile", action="store_true")
    ap.add_argument("--cpu-leg", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on a ONE-GPU box: gloo backend, every rank on cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the process group (RCCL; gloo with --rehearse) and run every barrier / all-reduce even at WORLD_SIZE 1: "
                         "the multi-GPU code path on a one-GPU box (launch with torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--param-checksum", action="store_true",
                    help="add `param_checksum` (wrap-around sum of the parameters' bit patterns after the timed steps) to the line")
    ap.add_argument("--verify-shard-step", action="store_true",
                    help="scene-shard: after the timed steps take one more step and recompute it from a snapshot with the dense "
                         "single-GPU Adam on the same reduced gradients; `shard_step_bit_exact` in the line")
    ap.add_argument("--sync-isect", action="store_true", help="read the intersection count back every step")
    ap.add_argument("--no-spatial-sort", action="store_true", help="A/B: keep the generator's (random) order of the Gaussians")
    ap.add_argument("--overlap-adam", default="after_binning", choices=["off", "after_project", "after_binning", "after_raster_fwd"],
                    help="Adam of the fully culled 64-Gaussian groups on a second stream (TrainConfig.overlap_culled_adam, as the CLI "
                         "runs it); off / other launch points for the A/B")
    ap.add_argument("--preset", default="splatfacto", choices=["splatfacto", "simple_trainer"],
                    help="which of the reference's two training jobs a step is: `ns-train splatfacto` (main.py:1270-1306, the pipeline's "
                         "default: absgrad, scale regulariser every 10 steps, random background, screen-size statistics) -- `value` -- or "
                         "gsplat's `simple_trainer default` (main.py:1318-1347).  The other preset is timed beside it (`presets`).")
    ap.add_argument("--one-preset", action="store_true", help="do not time the other preset beside the chosen one")
    ap.add_argument("--two-phase-binning", action="store_true",
                    help="A/B: mi3dgs_bin_count + mi3dgs_bin_emit instead of the fused mi3dgs_bin_tiles")
    return ap.parse_args()


def setup_dist(n_gpus, rehearse=False, force=False):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))     # RCCL
    elif n_gpus > 1:
        raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    return rank, world, local


def build_workload(args, rank, dev):
    from mi3dgs import scenes, trainer
    kw = {}
    if args.n:
        kw["n"] = args.n
    seed = {"cube": 0, "lego": 1, "garden": 2, "6m": 3}[args.scene] + (0 if args.mode == "scene-shard" else rank)
    log(f"generating scene {args.scene} seed {seed}")
    sc = scenes.make_scene(args.scene, seed=seed, **kw)
    # targets come from a perturbed copy of the SAME Gaussians: the loss is non-trivial but the
    # optimum is nearby, so Adam does not drive the scene (and the intersection count) away
    # from the named workload while it is being timed
    gp = torch.Generator().manual_seed(seed + 100)
    tgt = scenes.Scene(sc.name, {k: v.clone() for k, v in sc.params.items()}, sc.viewmats, sc.Ks, sc.width, sc.height)
    for k, sd in (("means", 0.01), ("scales", 0.1), ("opacities", 0.3), ("sh0", 0.1), ("shN", 0.02)):
        tgt.params[k] += sd * torch.randn(tgt.params[k].shape, generator=gp)
    V = min(args.views, sc.viewmats.shape[0])
    step = max(1, sc.viewmats.shape[0] // V)
    vidx = list(range(0, sc.viewmats.shape[0], step))[:V]
    g = sc.to(dev)
    vm, ks = g.viewmats[vidx].contiguous(), g.Ks[vidx].contiguous()
    # target images: rendered once by this engine from the *other* Gaussian set
    cfg0 = trainer.TrainConfig(densify=False)
    tr0 = trainer.Trainer(tgt.to(dev).params, vm, ks, torch.zeros(1, 1, 1, 3, device=dev), sc.width, sc.height, cfg0)
    imgs = []
    for i in range(V):
        imgs.append(tr0.render(vm[i], ks[i])[0].clamp(0, 1).clone())
        torch.cuda.synchronize()
        log(f"target view {i}: {int(tr0.last_binning['n_isect'].item())} intersections")
    imgs = torch.cat(imgs)
    del tr0
    torch.cuda.empty_cache()
    data = dict(params=g.params, vm=vm, ks=ks, imgs=imgs)
    tr = make_trainer(args, args.preset, sc, data, rank, dev)
    return sc, tr, V, data


def preset_config(args, preset, n):
    """TrainConfig of one of the reference's two training jobs, at the named workload's fixed size."""
    import dataclasses
    from mi3dgs import cli, trainer
    common = dict(
        max_steps=30_000, capacity=n + n // 8,
        # fixed-N workload: the densify statistics are accumulated every step (their cost is inside the
        # step); the every-100-steps refine pass is timed right behind the steps and amortised into `value`
        refine_start_iter=10 ** 9,
        max_isect=None if args.sync_isect else 0, fused_binning=not args.two_phase_binning, auto_isect_capacity=False,
        # the trainer's load-time Morton ordering of the Gaussians, as the CLI runs it (TrainConfig.spatial_sort_init)
        spatial_sort_init=not args.no_spatial_sort,
        overlap_culled_adam=None if args.overlap_adam == "off" else args.overlap_adam)
    if preset == "splatfacto":
        # `ns-train splatfacto --pipeline.model.use_scale_regularization=True` exactly as mi3dgs/cli.py configures it for the
        # reference's argv (main.py:1270-1306): absgrad, scale regulariser every 10 steps, random background, the screen-size
        # rules' statistic (step 3001 < stop_screen_size_at 4000) -- at the workload's full resolution (the coarse-to-fine
        # schedule has ended by step 6000 of 30 000)
        base = cli.splatfacto_config("splatfacto", 30_000, True, 8, n + n // 8)
        return dataclasses.replace(base, num_downscales=0, **common)
    return trainer.TrainConfig(**common)            # gsplat simple_trainer `default`


def make_trainer(args, preset, sc, data, rank, dev):
    from mi3dgs import trainer
    n = data["params"]["means"].shape[0]
    cfg = preset_config(args, preset, n)
    vm, ks, imgs = data["vm"], data["ks"], data["imgs"]
    if args.mode == "scene-shard":
        import dataclasses
        from mi3dgs import parallel
        world = int(os.environ.get("WORLD_SIZE", "1"))
        ctx = parallel.DistContext(rank, world, dev.index or 0)
        if args.force_dist and world == 1:
            # the one-GPU rehearsal of the multi-GPU path: a one-rank context that still takes every collective branch
            # (reduce-scatter -> Adam on the slice -> all-gather through the process group), not the fused single-GPU step
            class _OneRankActive(parallel.DistContext):
                @property
                def active(self):
                    return True
            ctx = _OneRankActive(rank, 1, dev.index or 0)
        tr = parallel.DataParallelTrainer(data["params"], vm, ks, imgs, sc.width, sc.height,
                                          dataclasses.replace(cfg, capacity=n, fuse_adam=not ctx.active), ctx=ctx)
    else:
        tr = trainer.Trainer(data["params"], vm, ks, imgs, sc.width, sc.height, cfg)
    tr.step_count = 3001          # SH degree 3 active (ramp finished); not a multiple of reset_every; every statistic still written
    if not args.sync_isect:
        # size the intersection buffers once (one sync here, none in the loop): 1.5x the worst view
        worst = 0
        tr.cfg.max_isect = None
        for i in range(vm.shape[0]):
            tr.render(vm[i], ks[i])
            worst = max(worst, int(tr.last_binning["n_isect"].item()) if hasattr(tr, "last_binning") else 0)
        tr.cfg.max_isect = int(worst * 1.5) + 1024
        log(f"[{preset}] intersection capacity {tr.cfg.max_isect} (worst view {worst})")
    return tr


def verify_shard_step(tr, view):
    """One step of the sharded optimiser (reduce-scatter -> Adam on this rank's rows with 1 / world -> all-gather) against the
    dense single-GPU update computed from a snapshot and the SAME summed gradients: bit for bit, every rank's verdict ANDed.
    (Two separate runs can not be compared bit for bit: rasterize_bwd's float atomics meet in a different order every launch.)"""
    import torch.distributed as dist
    from mi3dgs import ops
    from mi3dgs.trainer import GROUPS, WIDTHS
    m, c = tr.model, tr.cfg
    n = m.n
    tr._sync_optimizer_state()                   # every rank's moments current, as at a refine
    snap = {k: m.flat[k][m.cur].clone() for k in ("p", "m", "v")}
    lrs, step1 = tr.lrs(), tr.step_count + 1
    tr.step(view)
    tr._sync_optimizer_state()
    g = m.flat["g"].clone()                      # after the reduce-scatter: this rank's rows hold the sum over the ranks
    world = tr.ctx.world
    cap = m.capacity
    offs = [sum(WIDTHS[:i]) * cap for i in range(len(WIDTHS))]
    if world > 1:                                # the other ranks' rows of the summed gradient
        for _, off, cnt in tr._group_spans():
            L = cnt // world
            parts = [torch.empty(L, dtype=g.dtype, device=g.device) for _ in range(world)]
            dist.all_gather(parts, g[off + tr.ctx.rank * L: off + (tr.ctx.rank + 1) * L].contiguous())
            g[off: off + cnt] = torch.cat(parts)
    ops.adam_step([snap["p"][o: o + w * n] for o, w in zip(offs, WIDTHS)], [g[o: o + w * n] for o, w in zip(offs, WIDTHS)],
                  [snap["m"][o: o + w * n] for o, w in zip(offs, WIDTHS)], [snap["v"][o: o + w * n] for o, w in zip(offs, WIDTHS)],
                  lrs, step1, beta1=c.adam_beta1, beta2=c.adam_beta2, eps=c.adam_eps, grad_scale=1.0 / world)
    ok = all(torch.equal(snap[k][o: o + w * n], m.flat[k][m.cur][o: o + w * n]) for k in ("p", "m", "v") for o, w in zip(offs, WIDTHS))
    if dist.is_initialized():
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=g.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(int(t.item()))
    return bool(ok)


class StageTimer:
    def __init__(self):
        self.events = {}

    def __call__(self, name, thunk):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        thunk()
        e.record()
        self.events.setdefault(name, []).append((s, e))

    def summary(self, iters):
        out = {}
        for name, evs in self.events.items():
            tot = sum(s.elapsed_time(e) for s, e in evs)
            out[name] = dict(ms_per_step=tot / iters, calls_per_step=len(evs) / iters)
        return out


def cpu_baseline_cpu_tensors(P, vm, K, gt, W, H, crop_div=4):
    """One training step of the CPU oracle (a port: the reference has no CPU rasteriser) on a
    bounded sample, timed in three legs so that the extrapolation is explicit:
      A  project + SH forward over ALL Gaussians             (scales with N, not with the frame)
      B  binning + rasterise fwd + L1/SSIM + rasterise bwd on a centred (W/d x H/d) crop (d = 4: the 1/16 of SURVEY 8d)
      C  project + SH backward over ALL Gaussians
    full-frame step time = A + C + d*d * B.  Adam is not included (the oracle's is a 1-liner)."""
    from oracle import gs_oracle as O
    threads = host_threads()
    torch.set_num_threads(threads)
    cw, ch = W // crop_div, H // crop_div
    x0, y0 = (W - cw) // 2, (H - ch) // 2
    Kc = K.clone()
    Kc[0, 2] -= x0
    Kc[1, 2] -= y0
    gtc = gt[y0:y0 + ch, x0:x0 + cw][None]
    N = P["means"].shape[0]
    t0 = time.perf_counter()
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    op = torch.sigmoid(leaves["opacities"])
    radii, m2d, dep, con, _ = O.projection(leaves["means"], leaves["quats"], leaves["scales"].exp(), vm[None], Kc[None],
                                           cw, ch, opacities=op)
    campos = torch.linalg.inv(vm)[:3, 3]
    sh = torch.cat([leaves["sh0"], leaves["shN"]], dim=1)
    cols = torch.clamp(O.spherical_harmonics(3, leaves["means"] - campos, sh) + 0.5, min=0.0)[None]
    tA = time.perf_counter() - t0
    t0 = time.perf_counter()
    rin = [t.detach().requires_grad_(True) for t in (m2d, con, cols, op[None])]
    tw, th = math.ceil(cw / 16), math.ceil(ch / 16)
    _, ids, flat = O.isect_tiles(rin[0], radii, dep, 16, tw, th)
    offs = O.isect_offset_encode(ids, 1, tw, th)
    r, a, _ = O.rasterize_to_pixels(rin[0], rin[1], rin[2], rin[3], cw, ch, 16, offs, flat)
    loss = O.photometric_loss(r, gtc, 0.2)
    loss.backward()
    tB = time.perf_counter() - t0
    t0 = time.perf_counter()
    torch.autograd.backward([m2d, con, cols, op], [rin[0].grad, rin[1].grad, rin[2].grad, rin[3].grad[0]])
    tC = time.perf_counter() - t0
    # D: the Adam step over all six groups (the oracle's own update rule, torch.optim.Adam arithmetic, eps 1e-15)
    t0 = time.perf_counter()
    for k, v in leaves.items():
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        O.adam_step(v.detach(), g, torch.zeros_like(v), torch.zeros_like(v), 1, 1e-3)
    tD = time.perf_counter() - t0
    full = tA + tC + tD + crop_div * crop_div * tB
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=1.0 / full, unit="it/s", cores=threads, kind="port", cpu_model=cpu_model,
                sample=f"oracle/gs_oracle.py float32 on {threads} threads of {cpu_model}, one train step: project+SH over all {N} "
                       f"Gaussians fwd {tA:.2f} s + bwd {tC:.2f} s, Adam over all parameters {tD:.2f} s, and binning+rasterise+loss "
                       f"fwd/bwd on a {cw}x{ch} centre crop ({ids.numel()} intersections) {tB:.2f} s; extrapolated full frame "
                       f"= A + C + D + {crop_div**2} x B = {full:.1f} s")


def box_info() -> dict:
    """What kind of box this line was measured on ("slow box" has to be a recorded fact, not a label):
    partition modes, clock tables with the active level, power cap -- plain sysfs reads (amdgpu), taken before anything
    touches the GPU; no child process, no smi library."""
    import glob
    out = {"cpu": None, "gpus": []}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                out["cpu"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass

    def rd(path):
        try:
            return open(path).read().strip()
        except OSError:
            return None

    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if rd(os.path.join(dev, "vendor")) != "0x1002":
            continue
        e = {"card": os.path.basename(os.path.dirname(dev))}
        for k in ("current_compute_partition", "current_memory_partition", "mem_info_vram_total", "mem_busy_percent"):
            v = rd(os.path.join(dev, k))
            if v is not None:
                e[k] = v
        for k in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
            v = rd(os.path.join(dev, k))
            if v is not None:
                lv = [x.strip() for x in v.splitlines()]
                e[k] = {"levels": [x.rstrip(" *") for x in lv], "active": [x.rstrip(" *") for x in lv if x.endswith("*")]}
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            for k in ("power1_cap", "power1_cap_max", "power1_average", "power1_input", "freq1_input", "freq2_input"):
                v = rd(os.path.join(hw, k))
                if v is not None:
                    e[k] = v
        out["gpus"].append(e)
    # identical cards collapse into one entry with a count
    uniq = []
    for e in out["gpus"]:
        body = {k: v for k, v in e.items() if k not in ("card", "power1_average", "power1_input", "mem_busy_percent")}
        for u in uniq:
            if u["same"] == body:
                u["cards"].append(e["card"])
                break
        else:
            uniq.append({"same": body, "cards": [e["card"]]})
    out["gpus"] = [dict(u["same"], cards=u["cards"]) for u in uniq]
    return out


def hbm_yardstick(dev, n_read, n_write, mib_per_array=256, reps=5):
    """GB/s of the library's own 16-byte-per-lane stream (mi3dgs_debug_hbm_stream) on THIS device: n_read arrays read, their
    sum written to n_write arrays.  (1, 1) is the float4 copy MI355X_MICROARCH.md quotes at 6.29 TB/s."""
    from mi3dgs import _lib, ops
    fl = (mib_per_array << 20) // 4
    buf = torch.empty((n_read + n_write) * fl, dtype=torch.float32, device=dev).normal_()
    st = ops._stream(dev)
    _lib.call("mi3dgs_debug_hbm_stream", ops._p(buf), fl, n_read, n_write, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.call("mi3dgs_debug_hbm_stream", ops._p(buf), fl, n_read, n_write, st)
    e1.record()
    torch.cuda.synchronize()
    return round(reps * (n_read + n_write) * fl * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


def host_threads() -> int:
    """Threads the CPU leg may use: the process's CPU affinity, capped at the 16-core share a
    one-GPU box gets (os.cpu_count() reports the whole host and oversubscribes it)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(sc, tr, view, timeout_s=300):
    """Runs the CPU leg in a child process that never touches the GPU, under a wall-clock
    limit, so a slow host can not cost the GPU measurement."""
    import subprocess
    import tempfile
    m = tr.model
    blob = dict(P={g: m.p(g).detach().cpu() for g in ("means", "quats", "scales", "opacities", "sh0", "shN")},
                vm=tr.viewmats[view].detach().cpu(), K=tr.Ks[view].detach().cpu(),
                gt=tr.images[view].detach().cpu(), W=sc.width, H=sc.height)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "cpu_leg.pt")
        torch.save(blob, path)
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(host_threads()))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", path], env=env,
                                 capture_output=True, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return dict(value=None, unit="it/s", cores=host_threads(), kind="port",
                        sample=f"CPU leg exceeded its {timeout_s} s limit on this host and was stopped")
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return dict(value=None, unit="it/s", cores=host_threads(), kind="port",
                sample="CPU leg failed: " + out.stderr.strip()[-300:])


def cpu_leg_main(path):
    b = torch.load(path, weights_only=False)
    print(json.dumps(cpu_baseline_cpu_tensors(b["P"], b["vm"], b["K"], b["gt"], b["W"], b["H"])), flush=True)


def main():
    args = parse()
    if args.cpu_leg:
        return cpu_leg_main(args.cpu_leg)
    box = box_info()            # before anything touches the GPU
    rank, world, local = setup_dist(args.gpus, args.rehearse, args.force_dist)
    dist_on = world > 1 or args.force_dist          # collectives run (also at world 1 with --force-dist)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from mi3dgs import _lib, ops, trainer  # noqa: F401

    sc, tr, V, data = build_workload(args, rank, dev)

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            import torch.distributed as dist
            dist.barrier() if args.rehearse else dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist_on:
            import torch.distributed as dist
            t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.rehearse else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    shard = args.mode == "scene-shard"
    view_of = (lambda i: (i * world + rank) % V) if shard else (lambda i: i % V)      # one view per rank per step

    def time_steps(tr_, what):
        log(f"[{what}] warmup")
        for i in range(args.warmup):
            tr_.step(view_of(i))
        sync_all()
        log(f"[{what}] timed region")
        t0 = time.perf_counter()
        for i in range(args.steps):
            tr_.step(view_of(args.warmup + i))
        sync_all()
        dt_ = max_over_ranks(time.perf_counter() - t0)
        # a chained kernel that gave up waiting, or a tile list cut at the capacity, would have made the
        # timed steps wrong without failing them: the number is only published if the sticky word is clean
        bits = _lib.async_errors()
        if bits:
            raise SystemExit(f"bench: device error word {bits:#x} after the timed region (1/2: a chained kernel "
                             "gave up waiting, 4: tile intersections exceeded isect_capacity); no result published")
        log(f"[{what}] {world * args.steps / dt_:.2f} it/s without the refine pass")
        return dt_

    def time_refine(tr_, what):
        """One densify / prune pass at this size (every refine_every = 100 steps in training): decide + scan + scatter of all
        parameter and moment rows into the spare bank + the host sync on the new count.  Every rank runs it (collective in
        scene-shard mode); the slowest rank's time counts."""
        runs = []
        for rep in range(2):                 # the first call also pays the one-time set-up of a few tensor ops
            for i in range(3):               # fresh statistics for the pass
                tr_.step(view_of(i))
            sync_all()
            t2 = time.perf_counter()
            rinfo = tr_.refine(do_grow=True)
            torch.cuda.synchronize()
            runs.append((max_over_ranks(1e3 * (time.perf_counter() - t2)), rinfo))
        refine_ms, rinfo = runs[-1]
        log(f"[{what}] refine: {refine_ms:.2f} ms ({rinfo['n_before']} -> {rinfo['n_after']} Gaussians; first call {runs[0][0]:.1f} ms)")
        return dict(refine_ms=refine_ms, first_call_ms=runs[0][0], refine_every=100,
                    **{k: rinfo[k] for k in ("n_before", "n_after", "n_dup", "n_split", "n_prune")})

    def amortised(dt_, refine_):
        """it/s of K steps plus their share (K / 100) of a refine pass -- what a training run of this preset sustains."""
        return world * args.steps / (dt_ + args.steps / 100.0 * refine_["refine_ms"] * 1e-3)

    dt = time_steps(tr, args.preset)
    async_bits = 0
    shard_exact = None
    if args.verify_shard_step and shard:
        shard_exact = verify_shard_step(tr, view_of(args.warmup + args.steps))
        log(f"sharded step equals the dense Adam on the same gradients bit for bit: {shard_exact}")
    checksum = None
    if args.param_checksum:
        checksum = 0
        for g_ in trainer.GROUPS:
            checksum = (checksum + int(tr.model.p(g_).contiguous().view(torch.int32).to(torch.int64).sum().item())) & 0xFFFFFFFFFFFFFFFF

    # ---- render-only FPS (same scene, SH degree 3), untimed-region extra
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    nr = max(10, args.steps)
    for i in range(nr):
        tr.render(tr.viewmats[i % V], tr.Ks[i % V])
    torch.cuda.synchronize()
    fps = nr / (time.perf_counter() - t1)
    async_bits |= _lib.async_errors()
    if async_bits:
        raise SystemExit(f"bench: device error word {async_bits:#x} after the render loop; no result published")

    if rank == 0:
        n = tr.model.n
        Px = sc.width * sc.height
        n_isect = int(tr.last["binning"]["n_isect"].item())
        n_vis = int((tr.radii[:, :n] > 0).all(-1).sum().item())
        # Gaussians in aligned 64-groups with a visible member (what the fused backward + Adam handles when the culled groups'
        # update runs as a kernel of its own on the second stream: TrainConfig.overlap_culled_adam)
        vg = (tr.radii[0, : n // 64 * 64] > 0).all(-1).view(-1, 64).any(1)
        split_adam = bool(getattr(tr, "_overlap_on", False))
        n_vgrp = (int(vg.sum().item()) * 64 + n % 64) if split_adam else n
        stages, roof, render_roof = {}, None, None
        if shard and world > 1:
            args.no_stage_profile = True      # a step is collective in this mode: rank 0 can not run extra ones alone
        if not args.no_stage_profile:
            # per-kernel launch times: HIP events recorded by the library itself on the launch
            # stream around every kernel (include/mi3dgs.h: mi3dgs_profile_enable)
            iters = min(10, max(3, args.steps // 4))
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for i in range(iters):
                tr.step(i % V)
            prof = _lib.profile_read()
            _lib.profile_enable(False)
            # ALGORITHMIC bytes per launch (BASELINE.md section 3 / DESIGN.md section 4):
            # I intersections, n Gaussians, n_vis visible, Px pixels
            I = n_isect
            # bytes of a tile key: 2 when the library sorts 16-bit keys (the step does not read the keys back, every tile id
            # fits, and the sort is above the switch-over to the classic passes; MI3DGS_KEYS16=0 turns it off)
            tiles = ((sc.width + 15) // 16) * ((sc.height + 15) // 16)
            kb = 2 if (tiles <= 65536 and (tr.cfg.max_isect or 0) > (4 << 20) and not args.two_phase_binning
                       and os.environ.get("MI3DGS_KEYS16", "1") != "0") else 4
            alg = {
                "project_fwd": n * 44 + n_vis * (192 + 72),
                "tile_count": n * 8 + n_vis * 16 + n * 12,
                "rs_hist/depth": n * 4, "rs_scatter/depth": n * 16,
                "gather_tiles": n * 12,
                "tile_emit": n * 8 + n_vis * 24 + I * (4 + kb),
                "rs_hist/isect": I * kb, "rs_scatter/isect": I * (8 + 2 * kb),
                "tile_offsets": I * kb,
                "rasterize_fwd": I * 40 + Px * 20,
                "loss_fwd": Px * (24 + 36), "loss_bwd": Px * (60 + 12),
                "rasterize_bwd": I * 80 + Px * 32,
                "project_bwd": n * 44 + n_vis * (192 + 64 + 64) + n * 236,
                "adam": n * 1652,
                # fused backward + Adam: read p, m, v (708) and write them (708) for every Gaussian, plus
                # radii (8) and, for the visible ones, the splat and gradient records (128)
                # (with the culled groups split off: only the Gaussians of groups with a visible member, every radius still read)
                "project_bwd_adam": n_vgrp * (708 + 708) + n * 8 + n_vis * 128,
                # (the every-tenth step with splatfacto's scale regulariser: the same launch, its own tag)
                "project_bwd_adam/scale_reg": n_vgrp * (708 + 708) + n * 8 + n_vis * 128,
                # the culled groups' Adam on the second stream: p, m, v read and written, every radius read
                "adam_culled_groups": (n - n_vgrp) * (708 + 708) + n * 8,
            }
            for tag, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                e = dict(ms_per_step=ms / iters, launches_per_step=cnt / iters, us_per_launch=1e3 * ms / cnt)
                if tag in alg:
                    e["alg_bytes_per_launch"] = alg[tag]
                    e["alg_GBps"] = alg[tag] / (ms / cnt * 1e-3) / 1e9
                stages[tag] = e
            # pixel-splat pairs any implementation has to evaluate: every list entry up to the last
            # contributor of each pixel (SURVEY.md 8d "algorithmic flops": 20 / 60 flop per pair)
            lb = tr.last_binning
            lid = tr.raster_out["last_ids"][0].long()
            al = tr.raster_out["alphas"][0, ..., 0]
            offs = lb["isect_offsets"][0].long()
            tstart = offs.repeat_interleave(16, 0).repeat_interleave(16, 1)[: sc.height, : sc.width]
            pairs = int(((lid - tstart + 1) * (al > 0)).sum().item())
            flop = {"rasterize_fwd": 20.0 * pairs, "rasterize_bwd": 60.0 * pairs}
            for k, f in flop.items():
                if k in stages:
                    stages[k]["alg_flop_per_launch"] = f
                    stages[k]["alg_TFLOPs"] = f / (stages[k]["us_per_launch"] * 1e-6) / 1e12
            dom = max((k for k in stages if k in alg), key=lambda k: stages[k]["ms_per_step"])
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # measured offline with rocprofv3 --pmc
            pmc_tab = json.load(open(pmc)) if os.path.isfile(pmc) and args.scene == "garden" and not args.n else {}

            def offline_traffic(tag):
                # HBM bytes per launch from the committed counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
                # separate runs of this command, gfx950 correction applied; tools/profile_round.sh).  NOT measured
                # in this run: `traffic_source` says so.  Default workload only.
                k = tag.split("/")[0]
                return (pmc_tab.get(k) or pmc_tab.get(k + "_wave") or pmc_tab.get(k + "_mm") or {}).get("hbm_bytes_per_launch")

            traffic = offline_traffic(dom)
            common = dict(kernel=dom, traffic=traffic,
                          traffic_source="profiles/pmc_traffic.json (offline rocprofv3 --pmc passes of this command)" if traffic else None,
                          us_per_launch=stages[dom]["us_per_launch"],
                          launches_per_step=stages[dom]["launches_per_step"], alg_bytes_per_launch=alg[dom],
                          alg_GBps=stages[dom]["alg_GBps"], hbm_frac=stages[dom]["alg_GBps"] / HBM_PEAK_GBS)
            if dom in flop:
                # the rasteriser is bound by vector issue, not HBM (its gathers hit L2/MALL): price it
                # against the f32 matrix/vector pipe peak (f32 MFMA rate == f32 VALU rate on gfx950)
                ach = stages[dom]["alg_TFLOPs"]
                roof = dict(bound="mfma", achieved=ach, peak=F32_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / F32_PEAK_TFLOPS,
                            pairs=pairs, **common)
            else:
                ach = stages[dom]["alg_GBps"]
                roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, **common)
                # Beside the pin-rate peak: what the library's OWN float4 streams get from THIS box's HBM, measured live (the
                # guide measures 6.29 TB/s for the copy): the plain 1 R : 1 W copy and the dominant kernel's own 5 R : 4 W mix
                # (it reads 1.78 GB and writes 1.45 GB per launch).  frac_of_own_stream = the kernel against that mix.
                try:
                    roof["copy_GBps_this_box"] = hbm_yardstick(dev, 1, 1)
                    roof["mix_5r4w_GBps_this_box"] = hbm_yardstick(dev, 5, 4, mib_per_array=128)
                    roof["frac_of_own_stream"] = round(ach / roof["mix_5r4w_GBps_this_box"], 3)
                except Exception as e:      # out of memory on a crowded card: the fields are informational
                    roof["copy_GBps_this_box"] = None
                    log(f"copy probe skipped: {e}")
            # ---- the render path's dominant kernel, same accounting (its own profiled launches)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for i in range(iters):
                tr.render(tr.viewmats[i % V], tr.Ks[i % V])
            rprof = _lib.profile_read()
            _lib.profile_enable(False)
            rdom = max((k for k in rprof if k in alg), key=lambda k: rprof[k][1])
            rus = 1e3 * rprof[rdom][1] / rprof[rdom][0]
            lb = tr.last_binning
            lid = tr.raster_out["last_ids"][0].long()
            al = tr.raster_out["alphas"][0, ..., 0]
            offs = lb["isect_offsets"][0].long()
            tstart = offs.repeat_interleave(16, 0).repeat_interleave(16, 1)[: sc.height, : sc.width]
            rpairs = int(((lid - tstart + 1) * (al > 0)).sum().item())
            if rdom == "rasterize_fwd":
                ach = 20.0 * rpairs / (rus * 1e-6) / 1e12
                render_roof = dict(kernel=rdom, bound="mfma", achieved=ach, peak=F32_PEAK_TFLOPS, unit="TFLOP/s",
                                   frac=ach / F32_PEAK_TFLOPS, pairs=rpairs, us_per_launch=rus,
                                   traffic=offline_traffic(rdom))
            else:
                ach = alg[rdom] / (rus * 1e-6) / 1e9
                render_roof = dict(kernel=rdom, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                                   frac=ach / HBM_PEAK_GBS, us_per_launch=rus, alg_bytes_per_launch=alg[rdom],
                                   traffic=offline_traffic(rdom))
            render_roof["ms_per_frame_kernels"] = sum(v[1] for v in rprof.values()) / iters
        cpu = None
        log("stage profile done; cpu baseline")
        if not args.no_cpu_baseline and world == 1:        # the CPU leg runs at N = 1 only
            cpu = cpu_baseline(sc, tr, 0)
            log(f"cpu baseline: {cpu.get('value')}")
        xgmi = tr.xgmi_bytes_per_step() if shard else 0
        isect_cap = tr.cfg.max_isect
    # ---- every rank: the refine pass of this preset, then the reference's OTHER training job on the same data
    refine = time_refine(tr, args.preset)
    its = amortised(dt, refine)
    presets = {args.preset: dict(it_per_s=its, it_per_s_without_refine=world * args.steps / dt, ms_per_step_without_refine=1e3 * dt / args.steps,
                                 refine=refine, reference_job=PRESET_JOBS[args.preset])}
    if not args.one_preset:
        other = "simple_trainer" if args.preset == "splatfacto" else "splatfacto"
        del tr
        torch.cuda.empty_cache()
        tr2 = make_trainer(args, other, sc, data, rank, dev)
        dt2 = time_steps(tr2, other)
        refine2 = time_refine(tr2, other)
        async_bits |= _lib.async_errors()
        presets[other] = dict(it_per_s=amortised(dt2, refine2), it_per_s_without_refine=world * args.steps / dt2,
                              ms_per_step_without_refine=1e3 * dt2 / args.steps, refine=refine2, reference_job=PRESET_JOBS[other])
        del tr2
    if async_bits:
        raise SystemExit(f"bench: device error word {async_bits:#x} after the refine passes; no result published")
    if rank == 0:
        par = f"scene-per-gpu x{world}"
        what = "one independent scene per GPU"
        if shard:
            par = f"one scene, view-per-rank x{world}, sharded optimiser (reduce-scatter / Adam on 1/{world} / all-gather)"
            what = f"ONE scene on {world} GPUs, one view per rank per step; value counts training VIEWS per second"
        result = {
            "metric": "3DGS training iterations/s @1080p (render FPS reported alongside)",
            "value": its, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * world / its, "higher_is_better": True, "scaling": "strong" if shard else "weak",
            "value_note": f"{args.steps} timed steps of the `{args.preset}` preset plus their share ({args.steps}/100) of the every-100-steps "
                          "densify / prune pass, which is timed right behind them (`refine`): the rate a training run sustains.  "
                          "`presets` has both of the reference's training jobs on the same data, with and without that share.",
            "vs_baseline": None, "dtype": "f32",
            "dtype_note": "f32 arithmetic and accumulation throughout, except the transport of rasterize_bwd's per-splat pixel "
                          "sums to their f32 accumulators: two bf16 terms per addend (16-bit significand, <= 2^-16 relative, "
                          "unbiased; profiles/r03_bwd_terms_ab.txt: 5e-6 relative from the f32 sums per launch; profiles/r04_precision_ab.txt: over "
                          "30 000-step jobs the means of held-out PSNR differ by <= 0.04 dB from the all-f32 backward's, Gaussian counts by "
                          "<= 0.15 %, against a launch-to-launch sd of 0.3 dB); alpha evaluation uses three-term bf16 coefficients (24 bits, "
                          "exact products)",
            "data": "synthetic", "box": box,
            "config": {"workload": f"{sc.name}: {n} Gaussians, SH degree 3, {sc.width}x{sc.height}, {V} resident views, {what}",
                       "preset": args.preset, "reference_job": PRESET_JOBS[args.preset],
                       "gaussians": n, "visible": n_vis, "intersections": n_isect, "pixels": Px, "isect_capacity": isect_cap,
                       "parallelism": par, "mode": args.mode, "gaussians_in_morton_order": not args.no_spatial_sort,
                       "culled_groups_adam_on_second_stream": split_adam, "gaussians_in_groups_with_a_visible_member": n_vgrp,
                       "xgmi_bytes_per_rank_per_step": xgmi},
            "presets": presets, "process_group": (("gloo" if args.rehearse else "nccl") if dist_on else None),
            "param_checksum": checksum, "shard_step_bit_exact": shard_exact,
            "render_fps": fps, "roofline": roof, "render_roofline": render_roof, "refine": refine,
            "async_errors": async_bits, "cpu_baseline": cpu, "stages": stages,
        }
        print(json.dumps(result), flush=True)
    if dist_on:
        import torch.distributed as dist
        dist.barrier() if args.rehearse else dist.barrier(device_ids=[local])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
