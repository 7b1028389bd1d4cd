"""Generates tests/golden/*.pt from the float64 oracle (run here: `python oracle/make_golden.py`).

PARITY UNPINNED: these vectors come from this repository's own restatement, because the
reference ships neither the algorithm nor any golden data (see oracle/gs_oracle.py header).
They pin the oracle against accidental change and give the GPU tests fixed inputs/outputs.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import activated, small_scene  # noqa: E402
from oracle import gs_oracle as O  # noqa: E402


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    sc = small_scene(n=64, seed=123, big=True, width=32, height=32, n_views=2, fx=30.0)
    A = activated(sc.params)
    leaves = {k: v.clone().requires_grad_(True) for k, v in A.items()}
    bg = torch.tensor([[0.1, 0.2, 0.3], [0.9, 0.5, 0.0]], dtype=torch.float64)
    r, a, meta = O.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["sh"],
                                 sc.viewmats.double(), sc.Ks.double(), 32, 32, sh_degree=3, backgrounds=bg)
    gen = torch.Generator().manual_seed(99)
    gt = torch.rand(r.shape, generator=gen, dtype=torch.float64)
    loss = O.photometric_loss(r, gt, 0.2)
    loss.backward()
    blob = dict(
        inputs={k: v.clone() for k, v in sc.params.items()}, viewmats=sc.viewmats, Ks=sc.Ks, width=32, height=32,
        backgrounds=bg.float(), target=gt.float(),
        render=r.detach().float(), alphas=a.detach().float(), loss=float(loss.detach()),
        radii=meta["radii"], means2d=meta["means2d"].detach().float(), conics=meta["conics"].detach().float(),
        depths=meta["depths"].detach().float(), colors=meta["colors"].detach().float(),
        isect_ids=meta["isect_ids"], flatten_ids=meta["flatten_ids"], isect_offsets=meta["isect_offsets"],
        grads={k: v.grad.float() for k, v in leaves.items()},
    )
    torch.save(blob, os.path.join(out_dir, "tiny_scene.pt"))
    print("wrote", os.path.join(out_dir, "tiny_scene.pt"), "I =", meta["isect_ids"].numel(), "loss =", float(loss))


if __name__ == "__main__":
    main()
