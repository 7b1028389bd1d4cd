"""End-to-end run on a synthetic COLMAP dataset (there are no real datasets in this image):
ground-truth Gaussians -> rendered views + SfM-like sparse points on disk -> the `ns-train`
or `simple_trainer.py` shim -> held-out PSNR, Gaussian count, it/s.

  python tools/train_synthetic.py --gt 200000 --views 60 --width 960 --height 540 --steps 7000
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gt", type=int, default=200_000)
    ap.add_argument("--points", type=int, default=20_000)
    ap.add_argument("--views", type=int, default=60)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--steps", type=int, default=7000)
    ap.add_argument("--mode", default="simple_trainer", choices=["simple_trainer", "ns-train"])
    ap.add_argument("--strategy", default="default", choices=["default", "mcmc"])
    ap.add_argument("--keep", default=None, help="directory to keep the dataset and outputs in")
    ap.add_argument("--seed-noise", type=float, default=0.01, help="jitter of the SfM-like seed points")
    ap.add_argument("--model-big", action="store_true", help="ns-train mode: splatfacto-big instead of splatfacto")
    ap.add_argument("--random-colours", action="store_true", help="keep the generator's independent random colour per Gaussian")
    ap.add_argument("--backdrop", type=int, default=20_000,
                    help="Gaussians of an opaque shell (radius 25) around the scene, so that every pixel shows content like a "
                         "photograph (0 = none: uncovered pixels then carry the constant background 0.2, the round-1/2 dataset)")
    a, extra = ap.parse_known_args()         # extra: passed on to the shim (e.g. --opacity_reg 0.005)
    from PIL import Image
    from mi3dgs import cli, io_colmap, scenes, trainer
    dev = torch.device("cuda:0")
    root = a.keep or tempfile.mkdtemp(prefix="mi3dgs_synth_")
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    t0 = time.time()
    sc = scenes.make_garden_like(n=a.gt, seed=7, width=a.width, height=a.height, n_views=a.views,
                                 fx=1450.0 * a.width / 1920.0)
    sc.params["opacities"] += 1.5                          # a mostly opaque scene, like a trained one
    if not a.random_colours:
        # colours that vary smoothly in space, like surfaces do: the bench generator draws every Gaussian's colour independently,
        # a 200 k-splat noise field that no 20 k-point reconstruction can represent (default strategy stuck at 29 - 30 dB, the MCMC
        # strategy's photometric gradients average out and its opacity regulariser empties the scene: profiles/r03_train_30000.txt)
        m = sc.params["means"]
        rgb = 0.5 + 0.35 * torch.stack([torch.sin(1.3 * m[:, 0] + 0.5) * torch.cos(0.9 * m[:, 1]),
                                        torch.sin(1.1 * m[:, 1] + 1.0) * torch.cos(1.7 * m[:, 2] + 0.6 * m[:, 0]),
                                        torch.sin(0.7 * m[:, 0] - 1.2 * m[:, 1])], dim=-1)
        rgb = rgb + 0.04 * torch.randn(rgb.shape, generator=torch.Generator().manual_seed(5))
        sc.params["sh0"] = ((rgb - 0.5) / 0.28209479)[:, None, :].contiguous()
        sc.params["shN"] = sc.params["shN"] * 0.3
    sc.params["scales"] += np.log(2.5 * (2_000_000 / a.gt) ** (1 / 3))   # keep the surface covered at lower counts
    n_obj = a.gt
    if a.backdrop > 0:
        sc = scenes.add_backdrop(sc, a.backdrop, 25.0, (0.0, 0.0, 0.0))
    g = sc.to(dev)
    tr = trainer.Trainer(g.params, g.viewmats, g.Ks, torch.zeros(1, 1, 1, 3, device=dev), a.width, a.height,
                         trainer.TrainConfig(densify=False))
    cams = [io_colmap.Camera(1, "PINHOLE", a.width, a.height,
                             np.array([float(sc.Ks[0, 0, 0]), float(sc.Ks[0, 1, 1]), a.width / 2, a.height / 2]))]
    ims = []
    bg = torch.full((1, 3), 0.2, device=dev)
    for i in range(a.views):
        img = tr.render(g.viewmats[i], g.Ks[i], background=bg)[0][0].clamp(0, 1)
        Image.fromarray((img * 255).round().byte().cpu().numpy()).save(os.path.join(root, "images", f"f_{i:04d}.png"))
        V = sc.viewmats[i].double().numpy()
        ims.append(io_colmap.Image(i + 1, io_colmap.rotmat_to_qvec(V[:3, :3]), V[:3, 3].copy(), 1, f"f_{i:04d}.png"))
    n_all = sc.params["means"].shape[0]
    # SfM-like seeds: a.points of the scene's own Gaussians, plus the same fraction of the backdrop's
    sel = torch.randperm(n_obj, generator=torch.Generator().manual_seed(1))[: a.points]
    if n_all > n_obj:
        k = max(1, int(round(a.backdrop * a.points / n_obj)))
        sel = torch.cat([sel, n_obj + torch.randperm(n_all - n_obj, generator=torch.Generator().manual_seed(2))[:k]])
    a.points = sel.numel()
    xyz = (sc.params["means"][sel] + a.seed_noise * torch.randn(a.points, 3)).double().numpy()
    rgb = ((0.5 + 0.2820948 * sc.params["sh0"][sel, 0]).clamp(0, 1) * 255).byte().numpy()
    sparse = os.path.join(root, "sparse", "0") if a.mode == "simple_trainer" else os.path.join(root, "colmap", "sparse", "0")
    io_colmap.write_model(sparse, cams, ims, xyz, rgb)
    del tr
    torch.cuda.empty_cache()
    print(f"[synthetic] dataset in {root}: {a.views} views {a.width}x{a.height}, {a.points} points, "
          f"made in {time.time() - t0:.1f}s", flush=True)
    if a.mode == "simple_trainer":
        res = os.path.join(root, "exports")
        cli.main_simple_trainer([a.strategy, "--max_steps", str(a.steps), "--result-dir", res, "--data_factor", "1",
                                 "--steps_scaler", "1.0", "--disable_viewer", "--packed", "--batch-size", "1",
                                 "--data-dir", root] + extra)
        st = json.load(open(os.path.join(res, "stats", f"val_step{a.steps - 1:04d}.json")))
    else:
        os.chdir(root)
        model = "splatfacto-mcmc" if a.strategy == "mcmc" else ("splatfacto-big" if a.model_big else "splatfacto")
        cli.main_ns_train([model, "--timestamp", "train-stage-1", "--pipeline.model.use_scale_regularization=True",
                           "--max-num-iterations", str(a.steps)] + extra + ["colmap", "--data", root, "--downscale-factor", "1"])
        st = json.loads("".join(l for l in open("outputs/unnamed/splatfacto/train-stage-1/config.yml") if not l.startswith("#")))["stats"]
    print("[synthetic] result:", json.dumps(st), flush=True)


if __name__ == "__main__":
    main()
