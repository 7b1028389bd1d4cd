import sys, torch
sys.path.insert(0,'pipeline-pointcloud_amd'); sys.path.insert(0,'tests'); sys.path.insert(0,'.')
torch.set_num_threads(16)
from mi3dgs import scenes
from oracle import gs_oracle as O
from helpers import activated, crop_camera, rel_err
import test_gpu_configs as T
dev=torch.device("cuda:0")
sc=scenes.make_scene("garden")
A=activated(sc.params, torch.float32)
cam=0; x0,y0,cw,ch=880,560,160,96
Kc=crop_camera(sc.Ks[cam:cam+1],x0,y0); vm=sc.viewmats[cam:cam+1]
with torch.no_grad():
    radii=O.projection(A["means"].double(),A["quats"].double(),A["scales"].double(),vm.double(),Kc.double(),cw,ch,opacities=A["opacities"].double())[0]
idx=torch.nonzero((radii>0).all(-1)[0]).flatten()
g=torch.Generator().manual_seed(x0+y0)
wr=torch.randn(1,ch,cw,3,generator=g,dtype=torch.float64); wa=torch.randn(1,ch,cw,1,generator=g,dtype=torch.float64); bg=torch.rand(1,3,generator=g,dtype=torch.float64)
r_ref,a_ref,g_ref=T._oracle_fwd_bwd({k:v[idx] for k,v in A.items()},vm,Kc,cw,ch,wr,wa,bg=bg)
r,a,gr,meta=T._hip_fwd_bwd(A,vm,Kc,cw,ch,wr,wa,dev,bg=bg)
for k in ("means","quats","scales","opacities","sh"):
    print(k, rel_err(gr[k][idx], g_ref[k]))
d=(gr["quats"][idx].double()-g_ref["quats"]).norm(dim=1)
top=torch.topk(d,8).indices
print("total quats grad norm", float(g_ref["quats"].norm()))
for t in top.tolist():
    i=int(idx[t])
    print(i, "err",float(d[t]),"ref",g_ref["quats"][t].tolist(),"hip",gr["quats"][i].tolist(),"scales",A["scales"][i].tolist(),"radii",radii[0,i].tolist(), "quat", A["quats"][i].tolist())
# ---- which stage is off for the worst Gaussian?  oracle gradients w.r.t. the projected quantities
i = int(idx[top[0]])
vs = meta["v_splats"][0, i].cpu()
print("HIP radii", meta["radii"][0, i].tolist(), "oracle radii", radii[0, i].tolist())
Asub = {k: v[idx].double() for k, v in A.items()}
with torch.no_grad():
    rad, m2d, dep, con, _ = O.projection(Asub["means"], Asub["quats"], Asub["scales"], vm.double(), Kc.double(), cw, ch, opacities=Asub["opacities"])
    campos = torch.linalg.inv(vm.double())[:, :3, 3]
    cols = torch.clamp(O.spherical_harmonics(3, Asub["means"][None] - campos[:, None], Asub["sh"][None]) + 0.5, min=0)
m2d = m2d.clone().requires_grad_(True); con = con.clone().requires_grad_(True)
import math
tw, th = math.ceil(cw / 16), math.ceil(ch / 16)
_, ids, flat = O.isect_tiles(m2d, rad, dep, 16, tw, th)
offs = O.isect_offset_encode(ids, 1, tw, th)
rr, aa, _ = O.rasterize_to_pixels(m2d, con, cols, Asub["opacities"][None], cw, ch, 16, offs, flat, backgrounds=bg)
((rr * wr).sum() + (aa * wa).sum()).backward()
t = int(top[0])
print("oracle v_mean2d", m2d.grad[0, t].tolist(), "v_conic", con.grad[0, t].tolist())
print("HIP    v_mean2d", vs[0:2].tolist(), "v_conic", vs[2:5].tolist())
sp = meta["splats"][0, i].cpu()
print("HIP record xy", sp[0:2].tolist(), "conic", sp[2:5].tolist(), "opac", float(sp[5]), "rgb", sp[6:9].tolist(), "depth", float(sp[9]))
print("oracle     xy", m2d[0, t].tolist(), "conic", con[0, t].tolist(), "opac", float(Asub["opacities"][t]), "rgb", cols[0, t].tolist(), "depth", float(dep[0, t]))
# oracle rasteriser on HIP's own records (float64 copies): isolates the rasteriser from the projection
spv = meta["splats"][0][idx.to(meta["splats"].device)].cpu().double()
m2 = spv[None, :, 0:2].clone().requires_grad_(True); c2 = spv[None, :, 2:5].clone().requires_grad_(True)
col2 = spv[None, :, 6:9].clone().requires_grad_(True); o2 = spv[None, :, 5].clone().requires_grad_(True)
radh = meta["radii"][0][idx.to(meta["radii"].device)].cpu()[None]
_, ids2, flat2 = O.isect_tiles(m2, radh, spv[None, :, 9], 16, tw, th)
offs2 = O.isect_offset_encode(ids2, 1, tw, th)
rr2, aa2, _ = O.rasterize_to_pixels(m2, c2, col2, o2, cw, ch, 16, offs2, flat2, backgrounds=bg)
((rr2 * wr).sum() + (aa2 * wa).sum()).backward()
print("oracle-on-HIP-records v_mean2d", m2.grad[0, t].tolist(), "v_conic", c2.grad[0, t].tolist(), "v_opac", float(o2.grad[0, t]), "v_rgb", col2.grad[0, t].tolist())
print("HIP                   v_mean2d", vs[0:2].tolist(), "v_conic", vs[2:5].tolist(), "v_opac", float(vs[5]), "v_rgb", vs[6:9].tolist())
vsall = meta["v_splats"][0][idx.to(meta["v_splats"].device)].cpu().double()
for nm, a_, b_ in (("mean2d", vsall[:, 0:2], m2.grad[0]), ("conic", vsall[:, 2:5], c2.grad[0]), ("opac", vsall[:, 5], o2.grad[0]), ("rgb", vsall[:, 6:9], col2.grad[0])):
    print("rasteriser-only rel err", nm, rel_err(a_, b_))
