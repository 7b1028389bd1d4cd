"""Shared test inputs: small seeded scenes the CPU oracle finishes in seconds."""
import json
import math
import os
import time

import torch

from mi3dgs import scenes


def small_scene(n=400, seed=0, width=64, height=48, n_views=2, fx=60.0, big=False):
    """Random blob seen by a ring of cameras; scales large enough that splats cover many tiles."""
    g = torch.Generator().manual_seed(seed)
    means = torch.rand(n, 3, generator=g) * 2.0 - 1.0
    lo, hi = (0.05, 0.4) if big else (0.03, 0.2)
    ls = torch.rand(n, 3, generator=g) * (math.log(hi) - math.log(lo)) + math.log(lo)
    P = scenes._params(means, ls, g)
    P["shN"] = P["shN"] * 3.0
    vms, ks = scenes.ring_cameras(n_views, 4.0, 0.5, 2.0, fx, width, height, g)
    return scenes.Scene("test", P, vms, ks, width, height)


def activated(P, dtype=torch.float64):
    return dict(means=P["means"].to(dtype), quats=P["quats"].to(dtype), scales=P["scales"].to(dtype).exp(),
                opacities=torch.sigmoid(P["opacities"].to(dtype)),
                sh=torch.cat([P["sh0"], P["shN"]], dim=1).to(dtype))


def rel_err(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


# ---------------------------------------------------------------- reference-held real splat
def load_wolf():
    """The reference's own trained sample `source/Gradio/favorites/wolf.spz` (committed as the data
    fixture tests/golden/wolf.spz), decoded by the reference's own codec compiled from its sources
    (oracle/_ref/splat_converter, built by oracle/Makefile; the binary travels to the GPU box).
    Returns the 90 586 Gaussians in parameter space (checkpoint schema).  Opacity logits the SPZ
    decoder emits as +-inf (alpha 0 / 255) are clamped to +-12."""
    import os
    import shutil
    import subprocess
    import tempfile
    from mi3dgs import io_ply
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conv = os.path.join(root, "oracle", "_ref", "splat_converter")
    if not os.path.isfile(conv):
        import pytest
        pytest.skip("oracle/_ref/splat_converter not built (make -C oracle, needs /root/reference)")
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(os.path.join(root, "tests", "golden", "wolf.spz"), os.path.join(td, "wolf.spz"))
        subprocess.run([conv, os.path.join(td, "wolf.spz")], check=True, stdout=subprocess.DEVNULL)
        P = io_ply.read_ply(os.path.join(td, "wolf.ply"))
    P["opacities"] = torch.nan_to_num(P["opacities"], posinf=12.0, neginf=-12.0).clamp(-12.0, 12.0)
    return {k: v.float().contiguous() for k, v in P.items()}


def wolf_scene(n_views=3, width=96, height=64, fx=None, subsample=1):
    """The wolf seen by a small ring of cameras (it stands around (0.02, -0.53, 0), about 0.7 tall)."""
    P = load_wolf()
    if subsample > 1:
        P = {k: v[::subsample].contiguous() for k, v in P.items()}
    centre = torch.tensor([0.02, -0.53, 0.0])
    fx = fx if fx is not None else 1.1 * width
    vms, ks = [], []
    for i in range(n_views):
        a = 2.0 * math.pi * (i + 0.25) / n_views
        eye = centre + torch.tensor([1.0 * math.cos(a), 0.35, 1.0 * math.sin(a)])
        vms.append(scenes.look_at(eye, centre, up=(0.0, -1.0, 0.0)))
        ks.append(scenes._intrinsics(fx, width, height))
    return scenes.Scene("wolf", P, torch.stack(vms), torch.stack(ks), width, height)


def crop_camera(K, x0, y0):
    """Intrinsics of the window whose top-left pixel is (x0, y0) of the full frame."""
    Kc = K.clone()
    Kc[..., 0, 2] -= x0
    Kc[..., 1, 2] -= y0
    return Kc


# ------------------------------------------------- integer reference of the binning, any device
def isect_reference(radii, splats, tile_size, tw, th):
    """gsplat's isect_tiles + 64-bit stable sort + offsets restated with torch ops on the tensors'
    own device (the GPU at full BASELINE sizes): bounding-box tile lists, key = (camera*tiles+tile)
    << 32 | depth bits, ties in Gaussian-index order.  Returns tiles_per_gauss, isect_ids,
    flatten_ids, offsets."""
    C, N = radii.shape[:2]
    dev = radii.device
    ts = float(tile_size)
    m, r = splats[..., 0:2].float(), radii.float()
    x0 = torch.clamp(torch.floor((m[..., 0] - r[..., 0]) / ts), 0, tw).long()
    y0 = torch.clamp(torch.floor((m[..., 1] - r[..., 1]) / ts), 0, th).long()
    x1 = torch.clamp(torch.ceil((m[..., 0] + r[..., 0]) / ts), 0, tw).long()
    y1 = torch.clamp(torch.ceil((m[..., 1] + r[..., 1]) / ts), 0, th).long()
    vis = (radii > 0).all(-1)
    tiles = torch.where(vis, (x1 - x0) * (y1 - y0), torch.zeros_like(x0))
    cnt = tiles.flatten()
    sel = torch.nonzero(cnt > 0).flatten()
    c = cnt[sel]
    w = (x1 - x0).flatten()[sel]
    rep = torch.repeat_interleave(torch.arange(sel.numel(), device=dev), c)
    start = torch.cumsum(c, 0) - c
    local = torch.arange(int(c.sum()), device=dev) - start[rep]
    ty = y0.flatten()[sel][rep] + local // w[rep]
    tx = x0.flatten()[sel][rep] + local % w[rep]
    tid = (sel // N)[rep] * (tw * th) + ty * tw + tx
    dbits = splats[..., 9].contiguous().view(torch.int32).long().flatten() & 0xFFFFFFFF
    keys = (tid << 32) | dbits[sel][rep]
    order = torch.sort(keys, stable=True).indices
    keys, vals = keys[order], sel[rep][order]
    offs = torch.searchsorted((keys >> 32).contiguous(), torch.arange(C * tw * th, device=dev)).to(torch.int32)
    return tiles.to(torch.int32), keys, vals.to(torch.int32), offs.reshape(C, th, tw)


# ------------------------------------------------- self-describing failures (VERDICT r2 #1b)
EVIDENCE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "test_evidence")


def _evidence(name, info):
    """A failed bit-exact comparison says what differed: into the assertion message AND into a JSON file under
    gpurun_out/test_evidence/ (merged back from the GPU box by gpurun)."""
    try:
        from mi3dgs import _lib
        info["async_errors"] = _lib.async_errors(reset=False)
    except Exception as e:                       # never let the reporting hide the failure
        info["async_errors"] = f"unreadable: {e}"
    info["name"] = name
    try:
        os.makedirs(EVIDENCE_DIR, exist_ok=True)
        path = os.path.join(EVIDENCE_DIR, f"{name.replace('/', '_').replace(' ', '_')}_{int(time.time())}.json")
        with open(path, "w") as f:
            json.dump(info, f, indent=1)
        info["evidence_file"] = path
    except OSError:
        pass
    return json.dumps(info)


def assert_clean(ops, what):
    """The device error word right after a call, not at the end of the test: a chained kernel whose bounded wait ran out
    (bits 1, 2) or a truncated list (bit 4) must not surface as a bare mismatch three assertions later."""
    bad = ops._lib.async_errors(reset=False)
    assert bad == 0, _evidence(what, dict(kind="device error word set", bits=bad))


def assert_same(name, got, ref, **ctx):
    """torch.equal that keeps its evidence: shapes, how many entries differ, the first differing index with the values
    around it on both sides, the device error word, and whatever context the caller passes (counts, modes)."""
    if got.shape == ref.shape and torch.equal(got, ref):
        return
    info = dict(kind="tensor mismatch", shape_got=list(got.shape), shape_ref=list(ref.shape),
                ctx={k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v)) for k, v in ctx.items()})
    if got.shape == ref.shape:
        g, r = got.flatten(), ref.flatten()
        diff = g != r
        idx = torch.nonzero(diff).flatten()
        first, last = int(idx[0]), int(idx[-1])
        lo, hi = max(0, first - 2), min(g.numel(), first + 6)
        info.update(n_entries=g.numel(), n_differing=int(diff.sum()), first_index=first, last_index=last,
                    got_around_first=g[lo:hi].tolist(), ref_around_first=r[lo:hi].tolist())
    raise AssertionError(_evidence(name, info))


def assert_count(name, got, expected, **ctx):
    if int(got) != int(expected):
        raise AssertionError(_evidence(name, dict(kind="count mismatch", got=int(got), expected=int(expected), delta=int(got) - int(expected),
                                                  ctx={k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v)) for k, v in ctx.items()})))


def seg_ctl(ws):
    """The segment workspace's control words (csrc/rasterize_mfma.h SEG_CTL_*): items booked by the forward, the backward's
    dedicated workers counted out, tiles the forward walked as segments."""
    c = ws[:12].view(torch.int32).cpu().tolist()
    return dict(items=c[0], out=c[1], heavy=c[2])


def assert_seg_clear(ws):
    """What every backward leaves: nothing booked, nobody counted out."""
    c = seg_ctl(ws)
    assert c["items"] == 0 and c["out"] == 0 and c["heavy"] == 0, c
