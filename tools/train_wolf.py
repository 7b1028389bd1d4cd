"""A real trained splat through the whole drop-in path.  NOT the reference's container and NOT a
dataset of photographs: the only real asset the reference ships, source/Gradio/favorites/wolf.spz (committed as the
data fixture tests/golden/wolf.spz, decoded by the reference's own codec oracle/_ref/splat_converter), is rendered
by this engine from a ring of cameras to 8-bit PNGs, a COLMAP model with a SUB-SAMPLED, jittered point cloud is
written next to them, and the `ns-train` / `ns-export` shims train from that to held-out PSNR: real anisotropic
Gaussians and real SH as the target, u8 images, densification growing N, export in the trainer's frame.

  python tools/train_wolf.py --steps 7000 [--views 60 --width 960 --height 720 --points 9000]
"""
import argparse
import json
import math
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pipeline-pointcloud_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=60)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--points", type=int, default=9000)
    ap.add_argument("--steps", type=int, default=7000)
    ap.add_argument("--model", default="splatfacto")
    ap.add_argument("--mode", default="ns-train", choices=["ns-train", "simple_trainer"],
                    help="simple_trainer: gsplat's launcher (main.py:1318-1347) with --strategy default|mcmc taken from --model")
    ap.add_argument("--backdrop", type=int, default=12_000,
                    help="Gaussians of an opaque, smoothly coloured shell around the wolf (radius 9 x its extent): every pixel shows "
                         "content, like a photograph in a room (0 = none: constant background 0.15, the round-2 dataset)")
    ap.add_argument("--raster-mode", type=int, default=1,
                    help="precision A/B (profiles/r04_precision_ab.txt): 1 = the product backward (pixel sums as two bf16 terms); from the "
                         "EXPERIMENTS library: 3 = all-f32 cross-lane reduce-scatter backward, 4 = three bf16 terms (no absgrad)")
    a, extra = ap.parse_known_args()         # extra: passed on to ns-train
    if a.raster_mode != 1:
        os.environ["MI3DGS_LIB"] = os.path.join(ROOT, "pipeline-pointcloud_amd", "mi3dgs", "libmi3dgs_exp.so")
    from PIL import Image
    from helpers import load_wolf
    from mi3dgs import cli, io_colmap, io_ply, scenes, trainer
    if a.raster_mode != 1:
        from mi3dgs import _lib, ops
        assert _lib.lib().mi3dgs_debug_set_raster_mode(a.raster_mode) == 0, _lib.lib().mi3dgs_last_error()
        if a.raster_mode == 3:               # that backward walks whole lists and knows nothing of the forward's checkpoints
            ops.raster_seg_workspace = lambda *args, **kw: None
    dev = torch.device("cuda:0")
    root = tempfile.mkdtemp(prefix="mi3dgs_wolf_")
    os.makedirs(os.path.join(root, "images"))
    P = load_wolf()
    n = P["means"].shape[0]
    centre = P["means"].median(0).values
    ext = float((P["means"] - centre).abs().quantile(0.99))
    n_obj = n
    if a.backdrop > 0:
        P = scenes.add_backdrop(scenes.Scene("wolf", P, None, None, a.width, a.height), a.backdrop, 9.0 * ext, tuple(centre.tolist())).params
        n = P["means"].shape[0]
    g = {k: v.to(dev) for k, v in P.items()}
    fx = 1.25 * a.width
    vms, ks = [], []
    for i in range(a.views):
        az = 2.0 * math.pi * i / a.views
        el = 0.25 + 0.35 * math.sin(3.0 * az)                          # wobbling ring: views from several heights
        r = 3.2 * ext
        eye = centre + torch.tensor([r * math.cos(az) * math.cos(el), -r * math.sin(el), r * math.sin(az) * math.cos(el)])
        vms.append(scenes.look_at(eye, centre, up=(0.0, -1.0, 0.0)))
        ks.append(scenes._intrinsics(fx, a.width, a.height))
    vms, ks = torch.stack(vms), torch.stack(ks)
    tr = trainer.Trainer(g, vms.to(dev), ks.to(dev), torch.zeros(1, 1, 1, 3, device=dev), a.width, a.height,
                         trainer.TrainConfig(densify=False))
    bg = torch.full((1, 3), 0.15, device=dev)
    cams = [io_colmap.Camera(1, "PINHOLE", a.width, a.height, np.array([fx, fx, a.width / 2, a.height / 2]))]
    ims, cover = [], []
    for i in range(a.views):
        img, al = tr.render(vms[i].to(dev), ks[i].to(dev), background=bg)
        cover.append(float((al > 0.5).float().mean()))
        Image.fromarray((img[0].clamp(0, 1) * 255).round().byte().cpu().numpy()).save(os.path.join(root, "images", f"f_{i:04d}.png"))
        V = vms[i].double().numpy()
        ims.append(io_colmap.Image(i + 1, io_colmap.rotmat_to_qvec(V[:3, :3]), V[:3, 3].copy(), 1, f"f_{i:04d}.png"))
    sel = torch.randperm(n_obj, generator=torch.Generator().manual_seed(1))[: a.points]
    if n > n_obj:           # the same fraction of the backdrop's Gaussians as seeds
        k = max(1, int(round(a.backdrop * a.points / n_obj)))
        sel = torch.cat([sel, n_obj + torch.randperm(n - n_obj, generator=torch.Generator().manual_seed(2))[:k]])
    a.points = sel.numel()
    xyz = (P["means"][sel] + 0.002 * ext * torch.randn(a.points, 3)).double().numpy()
    rgb = ((0.5 + 0.2820948 * P["sh0"][sel, 0]).clamp(0, 1) * 255).byte().numpy()
    sparse = os.path.join(root, "colmap", "sparse", "0") if a.mode == "ns-train" else os.path.join(root, "sparse", "0")
    io_colmap.write_model(sparse, cams, ims, xyz, rgb)
    del tr
    torch.cuda.empty_cache()
    print(f"[wolf] target: {n} Gaussians of the reference's wolf.spz (real SH, extent {ext:.3f}) + backdrop; {a.views} views "
          f"{a.width}x{a.height} u8 PNG, object covers {100 * sum(cover) / len(cover):.0f} % of a frame; "
          f"{a.points} of {n} points (jittered) as the SfM cloud", flush=True)
    os.chdir(root)
    t0 = time.time()
    if a.mode == "simple_trainer":
        res = os.path.join(root, "exports")
        strat = "mcmc" if a.model.endswith("mcmc") else "default"
        cli.main_simple_trainer([strat, "--max_steps", str(a.steps), "--result-dir", res, "--data_factor", "1", "--steps_scaler", "1.0",
                                 "--disable_viewer", "--packed", "--batch-size", "1", "--data-dir", root] + extra)
        st = json.load(open(os.path.join(res, "stats", f"val_step{a.steps - 1:04d}.json")))
        print(f"[wolf] result: {json.dumps(st)} (simple_trainer.py {strat}) in {time.time() - t0:.1f}s wall", flush=True)
        return
    cli.main_ns_train([a.model, "--timestamp", "train-stage-1", "--viewer.quit-on-train-completion=True",
                       "--logging.local-writer.enable", "False", "--logging.profiler", "none",
                       "--pipeline.model.use_scale_regularization=True", "--max-num-iterations", str(a.steps)] + extra +
                      ["colmap", "--data", root, "--downscale-factor", "1"])
    base = os.path.join("outputs", "unnamed", "splatfacto", "train-stage-1")
    cli.main_ns_export(["gaussian-splat", "--load-config", os.path.join(base, "config.yml"), "--output-dir", os.path.join(root, "exports")])
    st = json.loads("".join(l for l in open(os.path.join(base, "config.yml")) if not l.startswith("#")))["stats"]
    out = io_ply.read_ply(os.path.join(root, "exports", "splat.ply"))
    print(f"[wolf] result: {json.dumps(st)}; exported {out['means'].shape[0]} Gaussians from {a.points} seed points "
          f"in {time.time() - t0:.1f}s wall (training frame: z-up, unit cube)", flush=True)


if __name__ == "__main__":
    main()
