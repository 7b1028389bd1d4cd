import sys, torch
sys.path.insert(0, "pipeline-pointcloud_amd"); sys.path.insert(0, "tests")
from mi3dgs import ops, scenes
dev = torch.device("cuda:0")
for kind, absgrad, with_bg in (("lego", True, True), ("garden", False, False)):
    sc = scenes.make_scene(kind)
    g = {k: v.to(dev) for k, v in sc.params.items()}
    W, H = sc.width, sc.height
    vm, K = sc.viewmats[1:2].to(dev).contiguous(), sc.Ks[1:2].to(dev).contiguous()
    radii, splats = ops.project_fwd(g["means"], g["quats"], g["scales"], g["opacities"], vm, K, W, H, sh0=g["sh0"], shN=g["shN"], sh_degree=3, flags=3)
    b = ops.bin_tiles(radii, splats, W, H, 16, tight=True)
    bg = torch.tensor([[0.3, 0.6, 0.1]], device=dev) if with_bg else None
    r, a, l = ops.rasterize_fwd(splats, b, W, H, 16, bg, {})
    gen = torch.Generator().manual_seed(9)
    vr = (torch.rand(1, H, W, 3, generator=gen) - 0.5).to(dev); va = (torch.rand(1, H, W, 1, generator=gen) - 0.5).to(dev)
    lib = ops._lib.lib(); outs = []
    for mode in (14, 3):
        lib.mi3dgs_debug_set_raster_mode(mode)
        outs.append(ops.rasterize_bwd(splats, b, W, H, a, l, vr, va, 16, bg, absgrad).clone())
    lib.mi3dgs_debug_set_raster_mode(1)
    x, y = outs
    n = 11 if absgrad else 9
    print(kind, "max rel col err", max(float((x[0, :, c].double() - y[0, :, c].double()).norm() / y[0, :, c].double().norm()) for c in range(n)), "finite", bool(torch.isfinite(x).all()), "errs", ops._lib.async_errors())
