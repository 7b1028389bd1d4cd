"""Fraction of a scene's Gaussians that at least one of the first G training views sees (projection cull of the library).
Input to the 'visible-only packing' estimate of DESIGN.md section 5.
    python tools/visible_union.py 6m 8"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
import torch          # noqa: E402


def main():
    from mi3dgs import ops, scenes
    kind, G = sys.argv[1], int(sys.argv[2])
    dev = torch.device("cuda:0")
    sc = scenes.make_scene(kind)
    g = {k: v.to(dev) for k, v in sc.params.items()}
    N = g["means"].shape[0]
    V = sc.viewmats.shape[0]
    out = {}
    for name, views in (("consecutive", list(range(G))), ("spread", [i * V // G for i in range(G)])):
        seen = torch.zeros(N, dtype=torch.bool, device=dev)
        per = []
        for c in views:
            radii, _ = ops.project_fwd(g["means"], g["quats"], g["scales"], g["opacities"], sc.viewmats[c:c + 1].to(dev).contiguous(),
                                       sc.Ks[c:c + 1].to(dev).contiguous(), sc.width, sc.height, sh0=g["sh0"], shN=g["shN"], sh_degree=3,
                                       flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC)
            vis = (radii > 0).all(-1)[0]
            per.append(round(float(vis.float().mean()), 4))
            seen |= vis
        out[name] = dict(views=views, visible_per_view=per, union=round(float(seen.float().mean()), 4))
    print(json.dumps(dict(scene=kind, n=N, n_views_total=V, **out)))


if __name__ == "__main__":
    main()
