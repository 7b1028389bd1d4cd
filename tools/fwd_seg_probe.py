"""Timeline of one rasterize_fwd_seg_kernel launch from a probe build (-DMI3DGS_OS_STAMPS): when its blocks started and ended.

  python tools/fwd_seg_probe.py --lib tools/ab/libmi3dgs_probe.so [--opacity-scale 0.1]
"""
import argparse
import ctypes as C
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    ap.add_argument("--opacity-scale", type=float, default=1.0)
    a = ap.parse_args()
    from mi3dgs import _lib, ops, scenes
    from helpers import load_wolf
    dev = torch.device("cuda:0")
    P = load_wolf()
    centre = P["means"].median(0).values
    ext = float((P["means"] - centre).abs().quantile(0.99))
    W, H = 960, 720
    P = scenes.add_backdrop(scenes.Scene("wolf", P, None, None, W, H), 12000, 9.0 * ext, tuple(centre.tolist())).params
    eye = centre + torch.tensor([3.2 * ext * math.cos(0.6) * math.cos(0.3), -3.2 * ext * math.sin(0.3), 3.2 * ext * math.sin(0.6) * math.cos(0.3)])
    vm = scenes.look_at(eye, centre, up=(0.0, -1.0, 0.0))[None].to(dev).contiguous()
    K = scenes._intrinsics(1.25 * W, W, H)[None].to(dev).contiguous()
    g = {k: v.to(dev) for k, v in P.items()}
    if a.opacity_scale != 1.0:
        g["opacities"] = torch.logit((torch.sigmoid(g["opacities"]) * a.opacity_scale).clamp(1e-4, 0.999))
    radii, splats = ops.project_fwd(g["means"], g["quats"], g["scales"], g["opacities"], vm, K, W, H, sh0=g["sh0"], shN=g["shN"],
                                    sh_degree=3, flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC)
    b = ops.bin_tiles(radii, splats, W, H, 16, tight=True, radii_in_records=True)
    bg = torch.tensor([[0.2, 0.3, 0.4]], device=dev)
    h = C.CDLL(os.path.abspath(a.lib))
    fn = h.mi3dgs_rasterize_fwd
    fn.restype, fn.argtypes = _lib._SIGNATURES["mi3dgs_rasterize_fwd"]
    ws = ops.raster_seg_workspace(b, 1, dev)
    r = torch.empty(1, H, W, 3, device=dev); al = torch.empty(1, H, W, 1, device=dev); last = torch.empty(1, H, W, dtype=torch.int32, device=dev)
    st = ops._stream(dev)
    for _ in range(5):
        rc = fn(1, W, H, 16, b["tile_width"], b["tile_height"], ops._p(splats), ops._p(b["isect_offsets"]), ops._p(b["flatten_ids"]),
                ops._p(b["n_isect"]), ops._p(bg), ops._p(r), ops._p(al), ops._p(last), ops._p(ws), ws.numel(), st)
        assert rc == 0
    torch.cuda.synchronize()
    buf = np.zeros((8192, 3), dtype=np.uint64)
    h.mi3dgs_debug_read_rf_stamps(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.nbytes))
    s = buf.astype(np.int64)
    nw = 2048
    used = s[:, 0] > 0
    t0 = s[used, 0].min()
    out = {}
    for name, sl in (("workers", slice(0, nw)), ("tiles", slice(nw, 8192))):
        q = s[sl]
        ok = (q[:, 0] > 0) & (q[:, 1] > 0)
        work = ok & (q[:, 2] > 0)
        st0, en = (q[:, 0] - t0) / 100.0, (q[:, 1] - t0) / 100.0
        out[name] = dict(blocks=int(ok.sum()), with_work=int(work.sum()),
                         start_us=dict(first=round(float(st0[ok].min()), 2), median=round(float(np.median(st0[ok])), 2), last=round(float(st0[ok].max()), 2)),
                         end_us_last=round(float(en[ok].max()), 2),
                         busy_duration_us=(dict(median=round(float(np.median((en - st0)[work])), 2), p90=round(float(np.percentile((en - st0)[work], 90)), 2),
                                                max=round(float((en - st0)[work].max()), 2)) if work.any() else None),
                         idle_duration_us_median=round(float(np.median((en - st0)[ok & ~work])), 2) if (ok & ~work).any() else None,
                         entries=dict(median=int(np.median(q[work, 2])), max=int(q[work, 2].max())) if work.any() else None)
    print(json.dumps(dict(n_isect=int(b["n_isect"].item()), **out), indent=1))


if __name__ == "__main__":
    main()
