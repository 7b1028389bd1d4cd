"""Stress of the fused binning call the training step makes (depth sort with dropped culled splats -> chained tile emit ->
16-bit tile sort -> offsets), the call whose result differed ONCE in test_full_size_binning...[garden-0] (round 2: output
lost; round 3, first suite run: n_isect 16 085 932 instead of 15 980 980).  Runs it `--iters` times on the S2 scene and checks
every result against the first one that agrees with the exact two-phase path; a deviating call is described (how many list
entries differ, which splat ids are duplicated / missing in the depth-sorted list the emit consumed, the device error word).

  python tools/binning_stress.py --iters 400                       # the product library
  python tools/binning_stress.py --iters 400 --lib tools/ab/libmi3dgs_r2.so   # the round-2 build (same C-ABI for these calls)
"""
import argparse
import ctypes as C
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--scene", default="garden")
    ap.add_argument("--cam", type=int, default=0)
    a = ap.parse_args()
    from mi3dgs import _lib, ops, scenes
    if a.lib:          # another build of the library: bind only what this tool calls (the ABI number may differ)
        h = C.CDLL(os.path.abspath(a.lib))
        for name in ("mi3dgs_last_error", "mi3dgs_project_fwd", "mi3dgs_bin_workspace_bytes", "mi3dgs_bin_count", "mi3dgs_bin_emit",
                     "mi3dgs_bin_tiles", "mi3dgs_async_errors"):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = _lib._SIGNATURES[name]
        _lib._lib = h
    dev = torch.device("cuda:0")
    sc = scenes.make_scene(a.scene)
    W, H = sc.width, sc.height
    g = {k: v.to(dev) for k, v in sc.params.items()}
    vm, K = sc.viewmats[a.cam:a.cam + 1].to(dev).contiguous(), sc.Ks[a.cam:a.cam + 1].to(dev).contiguous()
    N = g["means"].shape[0]
    keys = torch.empty(1, N, dtype=torch.int32, device=dev)
    radii, splats = ops.project_fwd(g["means"], g["quats"], g["scales"], g["opacities"], vm, K, W, H, sh0=g["sh0"], shN=g["shN"],
                                    sh_degree=3, flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC, depth_keys=keys)
    exact = ops.bin_tiles(radii, splats, W, H, 16, tight=True, fused=False, radii_in_records=True)      # two-phase, exact size
    It = int(exact["n_isect"].item())
    cap = It + (It >> 2)
    ref_ids, ref_offs = exact["flatten_ids"][:It].clone(), exact["isect_offsets"].clone()
    bad = []
    for it in range(a.iters):
        b = ops.bin_tiles(radii, splats, W, H, 16, max_isect=cap, tight=True, fused=True, depth_keys=keys.clone(),
                          radii_in_records=True, want_tile_keys=False)
        n = int(b["n_isect"].item())
        ok = n == It and torch.equal(b["flatten_ids"][:It], ref_ids) and torch.equal(b["isect_offsets"], ref_offs)
        if not ok:
            e = dict(iter=it, n_isect=n, expected=It, async_errors=_lib.async_errors(reset=True))
            m = min(n, It)
            e["entries_differing_in_common_prefix"] = int((b["flatten_ids"][:m] != ref_ids[:m]).sum())
            cnt_got = torch.bincount(b["flatten_ids"][:n].long(), minlength=N)
            cnt_ref = torch.bincount(ref_ids.long(), minlength=N)
            d = cnt_got - cnt_ref
            e["splats_with_more_entries"] = int((d > 0).sum())
            e["splats_with_fewer_entries"] = int((d < 0).sum())
            e["splats_exactly_doubled"] = int(((cnt_got == 2 * cnt_ref) & (cnt_ref > 0)).sum())
            e["splats_vanished"] = int(((cnt_got == 0) & (cnt_ref > 0)).sum())
            bad.append(e)
    print(json.dumps(dict(lib=a.lib or "product", scene=a.scene, iters=a.iters, n_isect=It, deviating_calls=len(bad), first=bad[:5])), flush=True)


if __name__ == "__main__":
    main()
