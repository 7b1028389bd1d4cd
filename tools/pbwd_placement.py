"""project_bwd_adam on the S2 workload, timed with HIP events inside real training steps: one JSON line per process.
The kernel's time depends on the physical pages behind its ~25 streams, which differ from process to process on the SAME box
(profiles/r03_pbwd_placement.txt).  --history chooses what the process did to the device memory before the trainer's buffers
were allocated: `bench` = what bench.py does (render the targets with a first trainer, free it), `first` = nothing (the model
is the first thing allocated), `churn` = allocate 48 GB in 64 MB pieces and free every other one first.
    python tools/pbwd_placement.py --history first"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pipeline-pointcloud_amd")]
import torch          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--history", default="bench", choices=["bench", "first", "churn"])
    a = ap.parse_args()
    import bench
    from mi3dgs import _lib, scenes, trainer
    dev = torch.device("cuda:0")
    if a.history == "bench":
        sys.argv = [sys.argv[0], "--no-cpu-baseline", "--preset", "simple_trainer"]
        sc, tr, V, _ = bench.build_workload(bench.parse(), 0, dev)
    else:
        junk = []
        if a.history == "churn":
            junk = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(768)]
            junk = junk[::2]
            torch.cuda.empty_cache()
        sc = scenes.make_scene("garden", seed=2)
        n = sc.params["means"].shape[0]
        V = 8
        g = sc.to(dev)
        vidx = list(range(0, sc.viewmats.shape[0], sc.viewmats.shape[0] // V))[:V]
        vm, ks = g.viewmats[vidx].contiguous(), g.Ks[vidx].contiguous()
        cfg = trainer.TrainConfig(max_steps=30_000, capacity=n + n // 8, refine_start_iter=10 ** 9, max_isect=24_000_000)
        imgs = torch.rand(V, sc.height, sc.width, 3, device=dev)
        tr = trainer.Trainer(g.params, vm, ks, imgs, sc.width, sc.height, cfg)
        tr.step_count = 3001
        del junk
    ev = []

    def hook(name, thunk):
        if name != "mi3dgs_project_bwd_adam":
            return thunk()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        thunk()
        e1.record()
        ev.append((e0, e1))

    for i in range(5):
        tr.step(i % V)
    torch.cuda.synchronize()
    _lib.STAGE_HOOK = hook
    for i in range(a.steps):
        tr.step(i % V)
    torch.cuda.synchronize()
    _lib.STAGE_HOOK = None
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev[2:])
    print(json.dumps(dict(history=a.history, project_bwd_adam_us=round(t[len(t) // 2], 1), minmax=[round(t[0], 1), round(t[-1], 1)],
                          errors=_lib.async_errors())), flush=True)


if __name__ == "__main__":
    main()
