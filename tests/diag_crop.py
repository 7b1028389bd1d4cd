"""Diagnostic (not a test): where does the HIP path's gradient on a camera crop of a big scene part from the float64 oracle?
Runs the crop of tests/test_gpu_configs.py::test_oracle_on_camera_crops_of_the_full_scene, ranks the Gaussians by their share
of the error, and for the worst ones prints both sides stage by stage (records, rasteriser gradients, parameter gradients).
    python tests/diag_crop.py garden 0 880 560 160 96
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "pipeline-pointcloud_amd")]
from helpers import activated, crop_camera, rel_err          # noqa: E402
from oracle import gs_oracle as O                            # noqa: E402


def main():
    from mi3dgs import scenes
    import mi3dgs
    kind, cam = sys.argv[1], int(sys.argv[2])
    x0, y0, cw, ch = (int(v) for v in sys.argv[3:7])
    dev = torch.device("cuda:0")
    sc = scenes.make_scene(kind)
    A = activated(sc.params, torch.float32)
    Kc = crop_camera(sc.Ks[cam:cam + 1], x0, y0)
    vm = sc.viewmats[cam:cam + 1]
    with torch.no_grad():
        proj64 = O.projection(A["means"].double(), A["quats"].double(), A["scales"].double(), vm.double(), Kc.double(), cw, ch,
                              opacities=A["opacities"].double())
    idx = torch.nonzero((proj64[0] > 0).all(-1)[0]).flatten()
    g = torch.Generator().manual_seed(x0 + y0)
    wr = torch.randn(1, ch, cw, 3, generator=g, dtype=torch.float64)
    wa = torch.randn(1, ch, cw, 1, generator=g, dtype=torch.float64)
    bg = torch.rand(1, 3, generator=g, dtype=torch.float64)
    # HIP
    gl = {k: v.detach().float().to(dev).requires_grad_(True) for k, v in A.items()}
    r, a, meta = mi3dgs.rasterization(gl["means"], gl["quats"], gl["scales"], gl["opacities"], gl["sh"], vm.to(dev), Kc.to(dev), cw, ch,
                                      sh_degree=3, backgrounds=bg.float().to(dev))
    ((r * wr.float().to(dev)).sum() + (a * wa.float().to(dev)).sum()).backward()
    gr = {k: v.grad.cpu() for k, v in gl.items()}
    sp = meta["splats"][0].cpu()[idx]
    vs = meta["v_splats"][0].cpu()[idx].double()
    rad_h = meta["radii"][0].cpu()[idx]
    # oracle, float64, list order from the library's float32 depths
    leaves = {k: v[idx].clone().double().requires_grad_(True) for k, v in A.items()}
    orig = O.isect_tiles
    O.isect_tiles = (lambda m2d, rad, dep, *aa, **kk: orig(m2d, rad, sp[None, :, 9].double(), *aa, **kk))
    try:
        r_ref, a_ref, m = O.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["sh"], vm.double(),
                                          Kc.double(), cw, ch, sh_degree=3, backgrounds=bg)
        for k in ("means2d", "conics", "colors"):
            m[k].retain_grad()
        ((r_ref * wr).sum() + (a_ref * wa).sum()).backward()
    finally:
        O.isect_tiles = orig
    print("visible", idx.numel(), "radii differ on", int((rad_h != m["radii"][0]).any(-1).sum()), "image max diff",
          float((r.cpu().double() - r_ref).abs().max()))
    print("records: mean2d", rel_err(sp[:, 0:2], m["means2d"][0]), "conic", rel_err(sp[:, 2:5], m["conics"][0]), "colour",
          rel_err(sp[:, 6:9], m["colors"][0]))
    print("rasteriser gradients: mean2d", rel_err(vs[:, 0:2], m["means2d"].grad[0]), "conic", rel_err(vs[:, 2:5], m["conics"].grad[0]),
          "colour", rel_err(vs[:, 6:9], m["colors"].grad[0]))
    for k in ("means", "quats", "scales", "opacities", "sh"):
        d = gr[k][idx].double() - leaves[k].grad
        per = d.flatten(1).norm(dim=1) if d.dim() > 1 else d.abs()
        ref = leaves[k].grad.flatten(1).norm(dim=1) if d.dim() > 1 else leaves[k].grad.abs()
        top = per.topk(6)
        print(f"{k}: rel {rel_err(gr[k][idx], leaves[k].grad):.3e}; worst six carry {float((top.values ** 2).sum() / (per ** 2).sum()):.3f} of err^2")
        if k == "quats":
            for j in top.indices.tolist():
                print(f"  #{j} (gaussian {int(idx[j])}): |err| {float(per[j]):.3e} |ref| {float(ref[j]):.3e} share of |ref total| {float(ref[j] / ref.norm()):.3f}")
                print("     radii hip/ref", rad_h[j].tolist(), m["radii"][0][j].tolist(), "scales", [f"{v:.2e}" for v in A["scales"][idx[j]].tolist()],
                      "opacity", float(A["opacities"][idx[j]]), "depth", float(sp[j, 9]))
                print("     mean2d hip/ref", sp[j, 0:2].tolist(), m["means2d"][0][j].tolist())
                print("     conic  hip/ref", sp[j, 2:5].tolist(), m["conics"][0][j].tolist())
                print("     v_mean2d hip/ref", vs[j, 0:2].tolist(), m["means2d"].grad[0][j].tolist())
                print("     v_conic  hip/ref", vs[j, 2:5].tolist(), m["conics"].grad[0][j].tolist())
                print("     v_quats  hip/ref", gr["quats"][idx[j]].tolist(), leaves["quats"].grad[j].tolist())
                print("     v_scales hip/ref", gr["scales"][idx[j]].tolist(), leaves["scales"].grad[j].tolist())


if __name__ == "__main__":
    main()
