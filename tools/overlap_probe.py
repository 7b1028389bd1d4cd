"""Can an HBM-bound kernel hide under the binning chain?  Times the bench step with an extra stand-alone Adam launch (the
traffic of the invisible third of the Gaussians) (a) absent, (b) on a side stream between the projection and the binning,
(c) on the main stream at the same place."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
import torch
import bench
from mi3dgs import ops

class A: scene = "garden"; n = None; gaussians = None; views = 4; sync_isect = False; rehearse = False; two_phase_binning = False; placement_tuning = False; mode = "replicas"
sc, tr, V = bench.build_workload(A, 0, torch.device("cuda:0"))
dev = torch.device("cuda:0")
n_inv = 670_000
numel = n_inv * 59
p, g, m, v = (torch.zeros(numel, device=dev) for _ in range(4))
side = torch.cuda.Stream()
mode = {"v": 0}
orig = ops.bin_tiles
def patched(*a, **k):
    if mode["v"] == 1:
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            ops.adam_step([p], [g], [m], [v], [1e-3], 1)
    elif mode["v"] == 2:
        ops.adam_step([p], [g], [m], [v], [1e-3], 1)
    return orig(*a, **k)
ops.bin_tiles = patched
def run(k, steps=40):
    mode["v"] = k
    for i in range(5): tr.step(i % V)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(steps):
        tr.step(i % V)
        if k == 1: torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return (time.time() - t0) / steps * 1e3
for rep in range(2):
    for k, name in ((0, "no extra launch"), (1, "side stream"), (2, "main stream")):
        print(f"{name:16s} {run(k):.3f} ms/step")
