"""Reproduces the non-finite projection gradients found in the MCMC synthetic run (a needle-thin Gaussian 0.09 in front of the
camera, far off-screen, with an all-zero gradient record) and shows which inputs they depend on."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pipeline-pointcloud_amd"))
from mi3dgs import ops
dev = torch.device("cuda:0")
def run(tag, ls0=-17.504308700561523, sh_degree=3, two_cams=False, z_shift=0.0, vgrad=0.0):
    means = torch.tensor([[-0.5355250835418701, -0.0883798897266388, -1.2471290826797485]], device=dev)
    quats = torch.tensor([[1.1715995073318481, 1.1246978044509888, 0.7527013421058655, 0.2929958701133728]], device=dev)
    scales = torch.tensor([[ls0, -0.5044186115264893, -3.654362201690674]], device=dev)
    opac = torch.tensor([-3.7583377361297607], device=dev)
    sh0 = torch.zeros(1, 1, 3, device=dev); shN = torch.zeros(1, 15, 3, device=dev)
    vm = torch.tensor([[[0.07606703788042068, 0.9971022605895996, 0.0009645117679610848, 0.0006943568005226552], [-0.3021939992904663, 0.022131966426968575, 0.9529894590377808, -0.0068558864295482635], [0.9502065777778625, -0.07278256118297577, 0.3030018210411072, 0.9749422073364258 + z_shift], [0.0, 0.0, 0.0, 1.0]]], device=dev)
    K = torch.tensor([[[725.0, 0, 480.0], [0, 725.0, 270.0], [0, 0, 1]]], device=dev)
    if two_cams:
        vm, K = vm.repeat(2, 1, 1), K.repeat(2, 1, 1)
    C = vm.shape[0]
    radii, splats = ops.project_fwd(means, quats, scales, opac, vm, K, 960, 540, sh0=sh0, shN=shN, sh_degree=sh_degree, flags=3)
    v = torch.full((C, 1, 16), vgrad, device=dev)
    out = ops.project_bwd(means, quats, scales, opac, vm, K, 960, 540, radii, splats, v, sh0=sh0, shN=shN, color_mode=ops.COLOR_SH, sh_degree=sh_degree, flags=3)
    print(f"{tag:34s} radii {radii[0, 0].tolist()} depth {splats[0, 0, 9].item():.4f} conic {[round(x, 4) for x in splats[0, 0, 2:5].tolist()]} "
          + " ".join(f"{k}={t.flatten().tolist()[:3]}" for k, t in out.items() if k in ("v_means", "v_scales", "v_opacities", "v_sh0")))
run("as found")
run("sh_degree 0", sh_degree=0)
run("generic kernel (two cameras)", two_cams=True)
run("0.2 further from the camera", z_shift=0.2)
run("log-scale -12", ls0=-12.0)
run("log-scale -10", ls0=-10.0)
run("log-scale -8", ls0=-8.0)
run("as found, record gradient 1e-3", vgrad=1e-3)
