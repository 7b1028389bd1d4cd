"""Fraction of the aligned 64-Gaussian groups none of whose members a view sees, with the generator's order and with the Gaussians in
Morton order (TrainConfig.spatial_sort_init).    python tools/culled_groups.py garden"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
import torch          # noqa: E402


def main():
    from mi3dgs import ops, scenes
    from mi3dgs.trainer import morton_order
    kind = sys.argv[1]
    dev = torch.device("cuda:0")
    sc = scenes.make_scene(kind)
    out = {}
    for name in ("generator", "morton"):
        P = sc.params if name == "generator" else {k: v[morton_order(sc.params["means"])] for k, v in sc.params.items()}
        g = {k: v.to(dev) for k, v in P.items()}
        N = g["means"].shape[0]
        res = []
        for c in (0, 40, 90, 140):
            radii, _ = ops.project_fwd(g["means"], g["quats"], g["scales"], g["opacities"], sc.viewmats[c:c + 1].to(dev).contiguous(),
                                       sc.Ks[c:c + 1].to(dev).contiguous(), sc.width, sc.height, sh0=g["sh0"], shN=g["shN"], sh_degree=3,
                                       flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC)
            vis = (radii > 0).all(-1)[0]
            n64 = N // 64 * 64
            grp = vis[:n64].view(-1, 64)
            res.append(dict(view=c, visible=round(float(vis.float().mean()), 3), groups_fully_culled=round(float((~grp.any(1)).float().mean()), 3),
                            groups_fully_visible=round(float(grp.all(1).float().mean()), 3)))
        out[name] = res
    print(json.dumps(dict(scene=kind, **out), indent=1))


if __name__ == "__main__":
    main()
