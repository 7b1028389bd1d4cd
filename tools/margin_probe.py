import sys, os, math, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/pipeline-pointcloud_amd")
import test_gpu_configs as T
from helpers import rel_err
dev = torch.device("cuda:0")
ops = T._ops()
sc = T._scene("garden")
W, H = sc.width, sc.height
g, vm, K, radii, splats, keys = T._project(sc, dev, 0, want_keys=True)
tpg_r, ids_r, flat_r, offs_r = T.isect_reference(radii, splats, 16, math.ceil(W / 16), math.ceil(H / 16))
I = ids_r.numel()
bg = torch.tensor([[0.1, 0.6, 0.3]], device=dev)
gen = torch.Generator().manual_seed(5)
v_r = torch.randn(1, H, W, 3, generator=gen).to(dev)
v_a = torch.randn(1, H, W, 1, generator=gen).to(dev)
for rep in range(6):
    bf = ops.bin_tiles(radii, splats, W, H, 16, max_isect=I + 4096, tight=False, fused=True, depth_keys=keys.clone())
    bt = ops.bin_tiles(radii, splats, W, H, 16, max_isect=I + 4096, tight=True, fused=True, depth_keys=keys.clone(), radii_in_records=True)
    b16 = ops.bin_tiles(radii, splats, W, H, 16, max_isect=I + 4096, tight=True, fused=True, depth_keys=keys.clone(), radii_in_records=True, want_tile_keys=False)
    It = int(bt["n_isect"].item())
    same = torch.equal(b16["flatten_ids"][:It], bt["flatten_ids"][:It]) and torch.equal(b16["isect_offsets"], bt["isect_offsets"])
    okb = torch.equal(bf["flatten_ids"][:I], flat_r) and torch.equal(bf["isect_offsets"], offs_r)
    ob, ot = {}, {}
    r_b, a_b, l_b = ops.rasterize_fwd(splats, bf, W, H, 16, bg, ob)
    r_t, a_t, l_t = ops.rasterize_fwd(splats, bt, W, H, 16, bg, ot)
    vs_b = ops.rasterize_bwd(splats, bf, W, H, a_b, l_b, v_r, v_a, 16, bg)
    vs_t = ops.rasterize_bwd(splats, bt, W, H, a_t, l_t, v_r, v_a, 16, bg)
    print(rep, "k16 same:", same, "box lists exact:", okb, "render equal:", torch.equal(r_b, r_t), "grad rel_err tight vs box: %.3e" % rel_err(vs_t[..., :9], vs_b[..., :9]), "async", ops._lib.async_errors())
