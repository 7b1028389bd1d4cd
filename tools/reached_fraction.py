"""How much of the sorted tile lists does the rasteriser actually walk?  (measure before building depth slabs.)
reached = sum over tiles of (furthest last contributor of any pixel - tile start + 1) / intersections, for
  S2 (the bench scene), the reference's wolf.spz at 1080p, and the ~260 k-Gaussian regime of tools/train_synthetic.py."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pipeline-pointcloud_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def measure(name, P, vm, K, W, H, dev, bg=None):
    from mi3dgs import ops
    g = {k: v.to(dev) for k, v in P.items()}
    radii, splats = ops.project_fwd(g["means"], g["quats"], g["scales"], g["opacities"], vm.to(dev)[None].contiguous(), K.to(dev)[None].contiguous(),
                                    W, H, sh0=g["sh0"], shN=g["shN"], sh_degree=3, flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC)
    b = ops.bin_tiles(radii, splats, W, H, 16, tight=True, radii_in_records=True)
    r, a, last = ops.rasterize_fwd(splats, b, W, H, 16, bg, {})
    I = int(b["n_isect"].item())
    offs = b["isect_offsets"][0].long()
    th, tw = offs.shape
    ends = torch.cat([offs.flatten()[1:], torch.tensor([I], device=dev)]).view_as(offs)
    lid = last[0].long()
    hit = a[0, ..., 0] > 0
    lid = torch.where(hit, lid, torch.full_like(lid, -1))
    pad = torch.full((th * 16, tw * 16), -1, dtype=torch.long, device=dev)
    pad[:H, :W] = lid
    tmax = pad.view(th, 16, tw, 16).amax(dim=(1, 3))
    reached = torch.clamp(tmax - offs + 1, min=0)
    n_vis = int((radii > 0).all(-1).sum())
    out = dict(scene=name, gaussians=int(P["means"].shape[0]), visible=n_vis, width=W, height=H, intersections=I,
               reached=int(reached.sum()), reached_fraction=round(float(reached.sum()) / max(I, 1), 4),
               mean_list=round(I / (th * tw), 1), mean_reached=round(float(reached.float().mean()), 1),
               tiles_walked_to_the_end=round(float((reached >= (ends - offs)).float().mean()), 4),
               mean_alpha=round(float(a.mean()), 4))
    print(json.dumps(out), flush=True)
    return out


def main():
    from helpers import load_wolf
    from mi3dgs import scenes
    dev = torch.device("cuda:0")
    sc = scenes.make_scene("garden")
    measure("S2 garden-like (bench)", sc.params, sc.viewmats[0], sc.Ks[0], sc.width, sc.height, dev)
    # wolf at 1080p, with and without the backdrop of tools/train_wolf.py
    P = load_wolf()
    centre = P["means"].median(0).values
    ext = float((P["means"] - centre).abs().quantile(0.99))
    eye = centre + torch.tensor([3.2 * ext * math.cos(0.6) * math.cos(0.3), -3.2 * ext * math.sin(0.3), 3.2 * ext * math.sin(0.6) * math.cos(0.3)])
    vm = scenes.look_at(eye, centre, up=(0.0, -1.0, 0.0))
    K = scenes._intrinsics(1.25 * 1920, 1920, 1080)
    measure("wolf.spz 1080p", P, vm, K, 1920, 1080, dev)
    Pb = scenes.add_backdrop(scenes.Scene("wolf", P, None, None, 1920, 1080), 12000, 9.0 * ext, tuple(centre.tolist())).params
    measure("wolf.spz + backdrop 1080p", Pb, vm, K, 1920, 1080, dev)
    # the regime of the synthetic end-to-end run: ~260 k larger, mostly opaque Gaussians at 1080p
    s2 = scenes.make_garden_like(n=260_000, seed=7, width=1920, height=1080, n_views=4)
    s2.params["opacities"] += 1.5
    s2.params["scales"] += np.log(2.5 * (2_000_000 / 260_000) ** (1 / 3))
    measure("synthetic 260k (train_synthetic regime) 1080p", s2.params, s2.viewmats[0], s2.Ks[0], 1920, 1080, dev)
    s3 = scenes.add_backdrop(s2, 20000, 25.0)
    measure("synthetic 260k + backdrop 1080p", s3.params, s3.viewmats[0], s3.Ks[0], 1920, 1080, dev)


if __name__ == "__main__":
    main()
