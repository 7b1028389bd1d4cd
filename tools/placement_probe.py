"""Placement lab for the fused backward+Adam kernel: ONE process, one trainer, many candidate
allocations of its parameter / moment arrays, the kernel timed on each.
  python tools/placement_probe.py [trials]
Prints one line per candidate: what was re-allocated, how, and the kernel time."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
import torch  # noqa: E402


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    from mi3dgs import _lib, scenes, trainer
    from mi3dgs.trainer import GROUPS, WIDTHS
    dev = torch.device("cuda:0")
    n = 2_000_000
    sc = scenes.make_garden_like(n=n, seed=2, n_views=2)
    cfg = trainer.TrainConfig(max_steps=30000, capacity=n, refine_start_iter=10 ** 9, max_isect=24_000_000)
    params = {k: v.to(dev) for k, v in sc.params.items()}
    tr = trainer.Trainer(params, sc.viewmats.to(dev), sc.Ks.to(dev), None, sc.width, sc.height, cfg)
    tr.images = torch.cat([tr.render(tr.viewmats[i], tr.Ks[i])[0].clone() for i in range(2)])
    tr.step_count = 3001

    def time_kernel():
        for i in range(2):
            tr.step(i % 2)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for i in range(4):
            tr.step(i % 2)
        torch.cuda.synchronize()
        k = _lib.profile_read()["project_bwd_adam"]
        _lib.profile_enable(False)
        return round(k[1] / k[0] * 1e3, 1)

    tr.step_count = 0
    import time
    t0 = time.time()
    rep = tr.tune_placement(sweeps=int(os.environ.get("SWEEPS", "2")), log=print)
    print(json.dumps({"tune_seconds": round(time.time() - t0, 2), **{k: round(v, 1) for k, v in rep.items()}}), flush=True)
    tr.step_count = 3001
    print(json.dumps({"what": "after tuning", "us": time_kernel()}), flush=True)
    print(json.dumps({"what": "again", "us": time_kernel()}), flush=True)


if __name__ == "__main__":
    main()
