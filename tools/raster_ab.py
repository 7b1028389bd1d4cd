"""Same-box A/B of rasterize_bwd (and rasterize_fwd) between builds of the library: same scene, same lists, the kernels of
each build timed alternately with HIP events; results compared against the first build's.

  python tools/raster_ab.py --libs pipeline-pointcloud_amd/mi3dgs/libmi3dgs.so tools/ab/libmi3dgs_r2.so [--scene garden] [--absgrad]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--scene", default="garden")
    ap.add_argument("--cam", type=int, default=0)
    ap.add_argument("--absgrad", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--modes", nargs="*", type=int, default=None, help="raster mode per lib (experiments builds), default 1")
    ap.add_argument("--seg", nargs="*", type=int, default=None, help="1 = with the segment workspace, per lib (ABI >= 5)")
    ap.add_argument("--wolf-size", nargs=2, type=int, default=[960, 720], help="image size of the wolf scene")
    ap.add_argument("--opacity-scale", type=float, default=1.0, help="multiply every opacity: < 1 makes the lists be walked to their ends")
    a = ap.parse_args()
    from mi3dgs import _lib, ops, scenes
    dev = torch.device("cuda:0")
    if a.scene == "wolf":        # the real-training regime: the reference's wolf.spz + backdrop at 960 x 720 (tools/train_wolf.py)
        import math
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import load_wolf
        P = load_wolf()
        centre = P["means"].median(0).values
        ext = float((P["means"] - centre).abs().quantile(0.99))
        WW, WH = a.wolf_size
        P = scenes.add_backdrop(scenes.Scene("wolf", P, None, None, WW, WH), 12000, 9.0 * ext, tuple(centre.tolist())).params
        eye = centre + torch.tensor([3.2 * ext * math.cos(0.6) * math.cos(0.3), -3.2 * ext * math.sin(0.3), 3.2 * ext * math.sin(0.6) * math.cos(0.3)])
        sc = scenes.Scene("wolf", P, scenes.look_at(eye, centre, up=(0.0, -1.0, 0.0))[None], scenes._intrinsics(1.25 * WW, WW, WH)[None], WW, WH)
        a.cam = 0
    else:
        sc = scenes.make_scene(a.scene)
    W, H = sc.width, sc.height
    g = {k: v.to(dev) for k, v in sc.params.items()}
    if a.opacity_scale != 1.0:
        g["opacities"] = torch.logit((torch.sigmoid(g["opacities"]) * a.opacity_scale).clamp(1e-4, 0.999))
    vm, K = sc.viewmats[a.cam:a.cam + 1].to(dev).contiguous(), sc.Ks[a.cam:a.cam + 1].to(dev).contiguous()
    N = g["means"].shape[0]
    radii, splats = ops.project_fwd(g["means"], g["quats"], g["scales"], g["opacities"], vm, K, W, H, sh0=g["sh0"], shN=g["shN"],
                                    sh_degree=3, flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC)
    b = ops.bin_tiles(radii, splats, W, H, 16, tight=True, radii_in_records=True)
    bg = torch.tensor([[0.2, 0.3, 0.4]], device=dev)
    r, al, last = ops.rasterize_fwd(splats, b, W, H, 16, bg, {})
    gen = torch.Generator().manual_seed(3)
    vr = (torch.rand(1, H, W, 3, generator=gen) - 0.5).to(dev)
    va = (torch.rand(1, H, W, 1, generator=gen) - 0.5).to(dev)
    st = ops._stream(dev)
    handles = []
    for i, path in enumerate(a.libs):
        h = C.CDLL(os.path.abspath(path))
        for name in ("mi3dgs_rasterize_bwd", "mi3dgs_rasterize_fwd", "mi3dgs_debug_set_raster_mode", "mi3dgs_last_error"):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = _lib._SIGNATURES[name]
        h.mi3dgs_abi_version.restype = C.c_int
        h._abi = h.mi3dgs_abi_version()
        sig_b, sig_f = _lib._SIGNATURES["mi3dgs_rasterize_bwd"][1], _lib._SIGNATURES["mi3dgs_rasterize_fwd"][1]
        if h._abi < 5:           # before the segment workspace
            h.mi3dgs_rasterize_fwd.argtypes = sig_f[:-3] + [C.c_void_p]
            h.mi3dgs_rasterize_bwd.argtypes = sig_b[:-4] + [C.c_void_p]
        if h._abi < 4:           # round-2 builds: no n_gaussians argument either
            h.mi3dgs_rasterize_bwd.argtypes = sig_b[:-5] + [C.c_void_p]
        h._seg = None
        if a.seg and a.seg[i]:
            assert h._abi >= 5
            h._seg = ops.raster_seg_workspace(b, 1, dev)
        if a.modes:
            assert h.mi3dgs_debug_set_raster_mode(a.modes[i]) == 0, h.mi3dgs_last_error()
        handles.append(h)

    def bwd_args(h, out):
        head = (1, W, H, 16, b["tile_width"], b["tile_height"], ops._p(splats), ops._p(b["isect_offsets"]), ops._p(b["flatten_ids"]),
                ops._p(b["n_isect"]), ops._p(bg), ops._p(al), ops._p(last), ops._p(vr), ops._p(va), int(a.absgrad), ops._p(out))
        if h._abi >= 5:
            return head + (N, ops._p(h._r) if h._seg is not None else None, ops._p(h._seg), 0 if h._seg is None else h._seg.numel(), st)
        return head + ((N, st) if h._abi >= 4 else (st,))

    def fwd(h, o):
        rc = h.mi3dgs_rasterize_fwd(1, W, H, 16, b["tile_width"], b["tile_height"], ops._p(splats), ops._p(b["isect_offsets"]),
                                    ops._p(b["flatten_ids"]), ops._p(b["n_isect"]), ops._p(bg), ops._p(o[0]), ops._p(o[1]), ops._p(o[2]),
                                    *((ops._p(h._seg), 0 if h._seg is None else h._seg.numel(), st) if h._abi >= 5 else (st,)))
        assert rc == 0, h.mi3dgs_last_error()
        h._r = o[0]

    outs = [torch.zeros(1, N, 16, device=dev) for _ in handles]
    fo = [(torch.empty_like(r), torch.empty_like(al), torch.empty_like(last)) for _ in handles]
    tb = [[] for _ in handles]
    tf = [[] for _ in handles]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i, h in enumerate(handles):          # a forward of its own first: a segmented backward reads what it left
        if a.modes:
            assert h.mi3dgs_debug_set_raster_mode(a.modes[i]) == 0, h.mi3dgs_last_error()
        fwd(h, fo[i])
    for rep in range(a.reps + 2):
        for i, h in enumerate(handles):
            for fn, arg, acc in (("bwd", outs[i], tb[i]), (fwd, fo[i], tf[i])):
                if fn == "bwd":
                    arg.zero_()
                if a.modes:      # (the same library loaded twice is ONE handle with one mode word: set it for every launch)
                    assert h.mi3dgs_debug_set_raster_mode(a.modes[i]) == 0, h.mi3dgs_last_error()
                e0.record()
                if fn == "bwd":
                    rc = h.mi3dgs_rasterize_bwd(*bwd_args(h, arg))
                    assert rc == 0, h.mi3dgs_last_error()
                else:
                    fwd(h, arg)
                e1.record()
                e1.synchronize()
                if rep >= 2:
                    acc.append(e0.elapsed_time(e1) * 1e3)
    res = []
    ref = outs[0].double()
    for i, path in enumerate(a.libs):
        d = outs[i].double()
        cols = 11 if a.absgrad else 9
        err = float((d[..., :cols] - ref[..., :cols]).norm() / ref[..., :cols].norm())
        if handles[i]._seg is not None:
            res_items = int(handles[i]._seg[:4].view(torch.int32)[0].item())
        else:
            res_items = None
        res.append(dict(lib=path, mode=(a.modes[i] if a.modes else 1), seg_items=res_items, bwd_us_median=sorted(tb[i])[len(tb[i]) // 2], bwd_us_min=min(tb[i]),
                        fwd_us_median=sorted(tf[i])[len(tf[i]) // 2], rel_diff_vs_first=err,
                        fwd_equal_first=bool(torch.equal(fo[i][0], fo[0][0]) and torch.equal(fo[i][2], fo[0][2])),
                        fwd_max_abs_diff=float((fo[i][0] - fo[0][0]).abs().max()), alpha_max_abs_diff=float((fo[i][1] - fo[0][1]).abs().max()),
                        last_ids_differ=int((fo[i][2] != fo[0][2]).sum())))
    probe = None
    for i, h in enumerate(handles):          # a probe build among the libraries: duration of every tile's own block against its list length
        if not hasattr(h, "mi3dgs_debug_read_rb_stamps"):
            continue
        import numpy as np
        outs[i].zero_()
        h.mi3dgs_rasterize_bwd(*bwd_args(h, outs[i]))
        torch.cuda.synchronize()
        buf = np.zeros((16384, 8), dtype=np.uint64)
        h.mi3dgs_debug_read_rb_stamps(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.nbytes))
        nt = b["tile_width"] * b["tile_height"]
        nw = 512 if h._seg is not None else 0          # (SEG_WORKERS; 1 024 for workspaces sized beyond a million intersections)
        st = buf[nw:nw + nt].astype(np.int64)
        ok = (st[:, 1] > 0) & (st[:, 2] > 0)
        dur, ln = (st[ok, 1] - st[ok, 0]) / 100.0, st[ok, 2].astype(np.float64)
        A = np.stack([np.ones_like(ln), ln], 1)
        coef = np.linalg.lstsq(A, dur, rcond=None)[0]
        t0 = st[st[:, 0] > 0, 0].min()
        span = (st[ok, 1].max() - t0) / 100.0
        s0, e0 = (st[ok, 0] - t0) / 100.0, (st[ok, 1] - t0) / 100.0
        edges = np.linspace(0, span, 17)
        act = [float((np.minimum(e0, edges[j + 1]) - np.maximum(s0, edges[j])).clip(min=0).sum() / (edges[j + 1] - edges[j])) for j in range(16)]
        order = np.argsort(s0)
        probe_extra = dict(sum_block_us=round(float(dur.sum()), 0), ideal_span_us_at_1024_slots=round(float(dur.sum()) / 1024, 1),
                           active_blocks_per_sixteenth=[int(round(x)) for x in act],
                           mean_walk_of_first_and_last_quarter_started=[int(ln[order[: len(order) // 4]].mean()), int(ln[order[-len(order) // 4:]].mean())])
        # where a SHORT block's time goes (at most 64 walked entries: one group, one flush)
        sh = ok & (st[:, 2] <= 64) & (st[:, 3] > 0) & (st[:, 4] > 0) & (st[:, 5] > 0)
        if sh.any():
            q = st[sh]
            med = lambda x: round(float(np.median(x)) / 100.0, 2)
            probe_extra["short_block_phases_us_median"] = dict(blocks=int(sh.sum()), pixel_values_and_reductions=med(q[:, 6] - q[:, 0]), operands=med(q[:, 3] - q[:, 6]), list_and_records=med(q[:, 4] - q[:, 3]),
                                                                walk=med(q[:, 5] - q[:, 4]), flush=med(q[:, 1] - q[:, 5]), total=med(q[:, 1] - q[:, 0]))
        probe = dict(lib=a.libs[i], blocks_with_work=int(ok.sum()), **probe_extra, fit_us=dict(fixed=round(float(coef[0]), 2), per_1000_entries=round(float(coef[1] * 1000), 2)),
                     duration_us=dict(median=round(float(np.median(dur)), 1), p90=round(float(np.percentile(dur, 90)), 1), max=round(float(dur.max()), 1)),
                     walked_entries=dict(median=int(np.median(ln)), p90=int(np.percentile(ln, 90)), max=int(ln.max())), span_us=round(float(span), 1),
                     short_blocks_median_us=round(float(np.median(dur[ln <= 64])), 1) if (ln <= 64).any() else None)
    print(json.dumps(dict(scene=a.scene, absgrad=a.absgrad, n_isect=int(b["n_isect"].item()), results=res, probe=probe), indent=1), flush=True)


if __name__ == "__main__":
    main()
