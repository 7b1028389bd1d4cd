"""project_bwd_adam on the bench workload, timed with HIP events inside real training steps: one JSON line per process.
The kernel's time depends on the physical pages behind its ~25 streams, which differ from process to process on the SAME box
(profiles/r03_pbwd_placement.txt: six fresh processes in a row alternate 596 / 651 us).  Run it in several fresh processes.
    python tools/pbwd_placement.py [--steps 12]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pipeline-pointcloud_amd")]
import torch          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    import bench
    from mi3dgs import _lib
    sys.argv = [sys.argv[0], "--no-cpu-baseline"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    sc, tr, V = bench.build_workload(args, 0, dev)
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
    for i in range(2 * a.steps):
        tr.step(i % V)
    torch.cuda.synchronize()
    _lib.STAGE_HOOK = None
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev[2:])
    print(json.dumps(dict(project_bwd_adam_us=round(t[len(t) // 2], 1), minmax=[round(t[0], 1), round(t[-1], 1)],
                          errors=_lib.async_errors())), flush=True)


if __name__ == "__main__":
    main()
