"""Timings of the input-side kernels (k-NN of the SfM points, INTER_AREA downscale, u8 -> f32):
  python tools/bench_input_side.py [--points 1000000]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))

import torch  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    a = ap.parse_args()
    from mi3dgs import _lib, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    n = a.points
    out = {}
    clouds = {
        "uniform": torch.rand(n, 3, generator=g),
        "sfm_like": torch.cat([torch.randn(n * 6 // 10, 3, generator=g) * torch.tensor([8.0, 8.0, 0.05]),
                               torch.randn(n * 4 // 10 - 100, 3, generator=g) * 1.5,
                               torch.randn(100, 3, generator=g) * 500.0]),
    }
    for name, pts in clouds.items():
        pts = pts.to(dev)
        ms = timed(lambda: ops.knn(pts, 3))
        out[f"knn3_{name}_ms"] = round(ms, 3)
        out[f"knn3_{name}_Mpts_per_s"] = round(pts.shape[0] / ms / 1e3, 1)
    # brute force (what the torch plumbing did before), on a bounded slice
    pts = clouds["uniform"].to(dev)
    m = 8192
    ms = timed(lambda: torch.topk(torch.cdist(pts[:m], pts), 4, dim=1, largest=False), reps=2)
    out["torch_bruteforce_ms_extrapolated"] = round(ms * n / m, 1)
    _lib.profile_enable(True)
    ops.knn(pts, 3)
    torch.cuda.synchronize()
    out["knn_kernels_us"] = {k: round(v[1] * 1e3, 1) for k, v in _lib.profile_read().items()}
    _lib.profile_enable(False)
    img = (torch.rand(2160, 3840, 3, generator=g) * 255).to(torch.uint8).to(dev)
    for (h, w) in ((1080, 1920), (540, 960), (1000, 1777)):
        ms = timed(lambda: ops.image_downscale_area(img, h, w), reps=10)
        out[f"area_4k_to_{w}x{h}_ms"] = round(ms, 3)
        out[f"area_4k_to_{w}x{h}_GBps"] = round((img.numel() + h * w * 3) / ms / 1e6, 1)
    im2 = img[:1080, :1920].contiguous()
    o = torch.empty(1080, 1920, 3, device=dev)
    ms = timed(lambda: ops.image_u8_to_f32(im2, o), reps=50)
    out["u8_to_f32_1080p_us"] = round(ms * 1e3, 2)
    out["u8_to_f32_1080p_GBps"] = round(im2.numel() * 5 / ms / 1e6, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
