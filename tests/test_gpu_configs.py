"""GPU parity at the sizes BASELINE.json names (configs[0..4]; generators: SURVEY.md 8d).

  S0 cube      10 k, 4 x 800 x 800   whole config against the float64 oracle, forward and all gradient groups
  S1 lego-like 300 k, 800 x 800      |  full size: integer work (tile lists, 64-bit keys, offsets) bit for bit
  S2 garden    2 M, 1920 x 1080      |  against a torch.sort(stable=True) restatement on the GPU; the float64
  S3 6M        6 M, 1920 x 1080      |  oracle on camera CROPS of the same Gaussians against the HIP path run on
                                        the identical cropped camera; Adam against torch.optim.Adam; properties
                                        (sum tiles = I, depth order inside a tile, alpha range, finite gradients,
                                        clean device error word) on the full frame and on a whole training step.
The oracle is the checker only; every HIP call goes through the C-ABI (mi3dgs._lib).
"""
import math

import pytest
import torch

from helpers import (activated, assert_clean, assert_count, assert_same, assert_seg_clear, crop_camera, isect_reference, rel_err,
                     seg_ctl, wolf_scene)
from oracle import gs_oracle as O

pytestmark = pytest.mark.gpu

_SCENES = {}


def _scene(kind):
    from mi3dgs import scenes
    if kind not in _SCENES:
        _SCENES.clear()                      # one big scene in host memory at a time
        _SCENES[kind] = scenes.make_scene(kind)
    return _SCENES[kind]


def _ops():
    from mi3dgs import ops
    return ops


def _oracle_fwd_bwd(A, viewmats, Ks, W, H, wr, wa, sh_degree=3, bg=None, lists_from=None, mask_marginal=False):
    """float64 oracle, one camera at a time (bounds the autograd graph); returns renders, alphas and the
    gradients of sum(render * wr) + sum(alpha * wa) for the five parameter groups.

    The reference takes three kinds of DISCRETE decisions on float32 bits, and a float64 oracle can take each of them the other
    way without either side being wrong; on a 2 M-Gaussian frame every kind occurs.  None of them is excused per Gaussian:
      lists_from = (means2d [C,N,2], radii [C,N,2], depths [C,N]) float32, the library's own records: the oracle builds its
        tile lists (tile rectangle of a splat, order inside a tile) from these values instead of its float64 ones rounded;
      mask_marginal: pixels where some reached splat's `alpha < 1/255 -> skip` test is closer to its threshold than float32
        resolves (oracle/gs_oracle.py rasterize_to_pixels, `marginal`) get weight ZERO in the test's loss, on both sides; the
        mask [C,H,W] is returned as a fourth value and the caller runs the library with the same masked weights."""
    leaves = {k: v.clone().double().requires_grad_(True) for k, v in A.items()}
    rs, als, masks = [], [], []
    orig = O.isect_tiles
    try:
        for c in range(viewmats.shape[0]):
            if lists_from is not None:
                m32, r32, d32 = (t[c:c + 1] for t in lists_from)
                O.isect_tiles = (lambda m2d, rad, dep, *a, _m=m32, _r=r32, _d=d32, **k: orig(_m.to(m2d), _r.to(rad), _d.to(dep), *a, **k))
            marg = {} if mask_marginal else None
            r, a, _ = O.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["sh"],
                                      viewmats[c:c + 1].double(), Ks[c:c + 1].double(), W, H, sh_degree=sh_degree,
                                      backgrounds=None if bg is None else bg[c:c + 1].double(), marginal=marg)
            keep = 1.0 if marg is None else (~marg["alpha_skip"])[..., None].double()
            ((r * wr[c:c + 1] * keep).sum() + (a * wa[c:c + 1] * keep).sum()).backward()
            rs.append(r.detach())
            als.append(a.detach())
            if marg is not None:
                masks.append(marg["alpha_skip"])
    finally:
        O.isect_tiles = orig
    out = (torch.cat(rs), torch.cat(als), {k: v.grad for k, v in leaves.items()})
    return out + (torch.cat(masks),) if mask_marginal else out


def _hip_fwd_bwd(A, viewmats, Ks, W, H, wr, wa, dev, sh_degree=3, bg=None):
    import mi3dgs
    gl = {k: v.detach().float().to(dev).requires_grad_(True) for k, v in A.items()}
    r, a, meta = mi3dgs.rasterization(gl["means"], gl["quats"], gl["scales"], gl["opacities"], gl["sh"],
                                      viewmats.to(dev), Ks.to(dev), W, H, sh_degree=sh_degree,
                                      backgrounds=None if bg is None else bg.float().to(dev))
    ((r * wr.float().to(dev)).sum() + (a * wa.float().to(dev)).sum()).backward()
    return r.detach().cpu(), a.detach().cpu(), {k: v.grad.cpu() for k, v in gl.items()}, meta


def _check_images(r, a, r_ref, a_ref, mean_tol=3e-5, q_tol=1e-3, max_tol=3e-2):
    dr, da = (r.double() - r_ref).abs(), (a.double() - a_ref).abs()
    assert dr.mean() < mean_tol and da.mean() < mean_tol, (float(dr.mean()), float(da.mean()))
    assert torch.quantile(dr.flatten()[:: max(1, dr.numel() // 1_000_000)], 0.999) < q_tol
    # f32 against f64: single pixels may flip one 1/255 skip or one transmittance stop -- one splat's worth
    assert dr.max() < max_tol and da.max() < max_tol, (float(dr.max()), float(da.max()))
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0


# ------------------------------------------------------------------ configs[0]: S0, whole config
def test_s0_cube_whole_config_matches_oracle(dev):
    sc = _scene("cube")
    assert sc.params["means"].shape[0] == 10_000 and sc.viewmats.shape[0] == 4 and (sc.width, sc.height) == (800, 800)
    A = activated(sc.params)
    g = torch.Generator().manual_seed(11)
    wr = torch.randn(4, 800, 800, 3, generator=g, dtype=torch.float64)
    wa = torch.randn(4, 800, 800, 1, generator=g, dtype=torch.float64)
    bg = torch.rand(4, 3, generator=g, dtype=torch.float64)
    r_ref, a_ref, g_ref = _oracle_fwd_bwd(A, sc.viewmats, sc.Ks, 800, 800, wr, wa, bg=bg)
    r, a, gr, meta = _hip_fwd_bwd(A, sc.viewmats, sc.Ks, 800, 800, wr, wa, dev, bg=bg)
    _check_images(r, a, r_ref, a_ref)
    assert a_ref.max() > 0.9
    for k in ("means", "quats", "scales", "opacities", "sh"):
        e = rel_err(gr[k], g_ref[k])
        assert e < 2e-3, (k, e)
    assert _ops()._lib.async_errors() == 0


# ------------------------------------------ configs[1], [2], [4]: full-size integer work + properties
def _project(sc, dev, cam=0, want_keys=False, W=None, H=None, K=None):
    ops = _ops()
    g = {k: v.to(dev) for k, v in sc.params.items()}
    W, H = W or sc.width, H or sc.height
    vm = sc.viewmats[cam:cam + 1].to(dev).contiguous()
    K = (sc.Ks[cam:cam + 1] if K is None else K).to(dev).contiguous()
    keys = torch.empty(1, g["means"].shape[0], dtype=torch.int32, device=dev) if want_keys else None
    radii, splats = ops.project_fwd(g["means"], g["quats"], g["scales"], g["opacities"], vm, K, W, H, sh0=g["sh0"],
                                    shN=g["shN"], sh_degree=3, flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC,
                                    depth_keys=keys)
    return g, vm, K, radii, splats, keys


@pytest.mark.parametrize("kind,cam", [("lego", 7), ("garden", 0), ("6m", 3)])
def test_full_size_binning_bit_exact_and_raster_properties(dev, kind, cam):
    ops = _ops()
    sc = _scene(kind)
    W, H = sc.width, sc.height
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    g, vm, K, radii, splats, keys = _project(sc, dev, cam, want_keys=True)
    N = radii.shape[1]
    assert N == {"lego": 300_000, "garden": 2_000_000, "6m": 6_000_000}[kind]
    vis = (radii > 0).all(-1)
    assert torch.equal(keys[vis], splats[..., 9][vis].view(torch.int32)) and bool((keys[~vis] == -1).all())
    # (1) gsplat's bounding-box lists, exact allocation (two-phase path): bit for bit against the torch restatement
    tpg_r, ids_r, flat_r, offs_r = isect_reference(radii, splats, 16, tw, th)
    I = ids_r.numel()
    assert I > 10 * tw * th
    # (every comparison keeps its evidence on a mismatch -- count deltas, first differing index, the device error word -- and
    #  the error word is read after EVERY binning call: helpers.assert_same / assert_count / assert_clean)
    tag = f"full_size_binning/{kind}"
    b = ops.bin_tiles(radii, splats, W, H, 16, want_isect_ids=True, want_tiles_per_gauss=True, tight=False)
    assert_clean(ops, f"{tag}/two-phase box")
    assert_count(f"{tag}/two-phase box n_isect", b["n_isect"].item(), I, tiles_per_gauss_sum=int(b["tiles_per_gauss"].sum()))
    assert I == int(b["tiles_per_gauss"].sum())
    assert_same(f"{tag}/two-phase box tiles_per_gauss", b["tiles_per_gauss"], tpg_r)
    assert_same(f"{tag}/two-phase box isect_ids", b["isect_ids"], ids_r)
    assert_same(f"{tag}/two-phase box flatten_ids", b["flatten_ids"], flat_r)
    assert_same(f"{tag}/two-phase box isect_offsets", b["isect_offsets"], offs_r)
    # (2) the same lists from the fused capacity path (what the training step runs), keys from the projection
    bf = ops.bin_tiles(radii, splats, W, H, 16, max_isect=I + 4096, tight=False, fused=True, depth_keys=keys.clone(),
                       want_tiles_per_gauss=True)
    assert_clean(ops, f"{tag}/fused box")
    assert_count(f"{tag}/fused box n_isect", bf["n_isect"].item(), I, tiles_per_gauss_sum=int(bf["tiles_per_gauss"].sum()))
    assert_same(f"{tag}/fused box flatten_ids", bf["flatten_ids"][:I], flat_r)
    assert_same(f"{tag}/fused box isect_offsets", bf["isect_offsets"], offs_r)
    assert_same(f"{tag}/fused box tiles_per_gauss", bf["tiles_per_gauss"], tpg_r)
    # (3) exact ellipse culling: a subset of the box lists, same order inside every tile, identical render
    bt = ops.bin_tiles(radii, splats, W, H, 16, max_isect=I + 4096, tight=True, fused=True, depth_keys=keys.clone(),
                       want_tiles_per_gauss=True, radii_in_records=True)      # exactly the training step's call
    assert_clean(ops, f"{tag}/fused tight")
    It = int(bt["n_isect"].item())
    assert 0 < It <= I
    assert_count(f"{tag}/fused tight n_isect vs its own tiles_per_gauss", It, int(bt["tiles_per_gauss"].sum()), box_count=I)
    # the two-phase path (exact allocation, separate count and emit kernels, no dropped sentinels in the depth sort) gives the
    # same tight lists: an independent second route to every number below
    bt2 = ops.bin_tiles(radii, splats, W, H, 16, tight=True, fused=False, radii_in_records=True)
    assert_clean(ops, f"{tag}/two-phase tight")
    assert_count(f"{tag}/fused tight n_isect vs two-phase tight", It, bt2["n_isect"].item(), box_count=I)
    assert_same(f"{tag}/fused tight flatten_ids vs two-phase", bt["flatten_ids"][:It], bt2["flatten_ids"][:It], n_isect=It)
    assert_same(f"{tag}/fused tight isect_offsets vs two-phase", bt["isect_offsets"], bt2["isect_offsets"], n_isect=It)
    # the training step's and the renderer's call does not ask for the sorted tile keys back: the library then sorts 16-bit
    # keys (garden and 6m are above the switch-over to the classic passes, lego below it) -- the same lists bit for bit
    b16 = ops.bin_tiles(radii, splats, W, H, 16, max_isect=I + 4096, tight=True, fused=True, depth_keys=keys.clone(),
                        radii_in_records=True, want_tile_keys=False)
    assert_clean(ops, f"{tag}/fused tight 16-bit keys")
    assert b16["tile_keys"] is None
    assert_count(f"{tag}/fused tight 16-bit keys n_isect", b16["n_isect"].item(), It, box_count=I)
    assert_same(f"{tag}/fused tight 16-bit keys flatten_ids", b16["flatten_ids"][:It], bt["flatten_ids"][:It], n_isect=It)
    assert_same(f"{tag}/fused tight 16-bit keys isect_offsets", b16["isect_offsets"], bt["isect_offsets"], n_isect=It)
    tk, fi = bt["tile_keys"][:It].long(), bt["flatten_ids"][:It].long()
    assert bool((tk[1:] >= tk[:-1]).all())
    d = splats.view(-1, ops.SPLAT_STRIDE)[fi, 9]
    same = tk[1:] == tk[:-1]
    assert bool((d[1:][same] >= d[:-1][same]).all())                       # depth order inside every tile
    tie = same & (d[1:] == d[:-1])
    assert bool((fi[1:][tie] > fi[:-1][tie]).all())                        # ties in Gaussian-index order
    pair_t = tk * N + fi
    pair_b = ((ids_r >> 32) * N + flat_r.long()).sort().values
    pos = torch.searchsorted(pair_b, pair_t).clamp(max=I - 1)
    assert bool((pair_b[pos] == pair_t).all())                             # tight is a subset of box
    bg = torch.tensor([[0.1, 0.6, 0.3]], device=dev)
    out_b, out_t = {}, {}
    r_b, a_b, l_b = ops.rasterize_fwd(splats, bf, W, H, 16, bg, out_b)
    r_t, a_t, l_t = ops.rasterize_fwd(splats, bt, W, H, 16, bg, out_t)
    assert torch.equal(r_b, r_t) and torch.equal(a_b, a_t)                 # the dropped pairs never contribute
    assert float(a_t.min()) >= 0.0 and float(a_t.max()) <= 1.0 and bool(torch.isfinite(r_t).all())
    # every pixel's last contributor lies inside its tile's list
    offs = bt["isect_offsets"][0].long()
    ends = torch.cat([offs.flatten()[1:], torch.tensor([It], device=dev)]).view_as(offs)
    up = lambda t: t.repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]   # noqa: E731
    hit = a_t[0, ..., 0] > 0
    assert bool(((l_t[0].long() >= up(offs)) & (l_t[0].long() < up(ends)))[hit].all())
    # (4) backward on both lists: same gradients up to float-atomic order, finite
    gen = torch.Generator().manual_seed(5)
    v_r = torch.randn(1, H, W, 3, generator=gen).to(dev)
    v_a = torch.randn(1, H, W, 1, generator=gen).to(dev)
    vs_b = ops.rasterize_bwd(splats, bf, W, H, a_b, l_b, v_r, v_a, 16, bg)
    vs_t = ops.rasterize_bwd(splats, bt, W, H, a_t, l_t, v_r, v_a, 16, bg)
    assert bool(torch.isfinite(vs_t).all())
    assert rel_err(vs_t[..., :9], vs_b[..., :9]) < 1e-5
    assert float(vs_t[0][~vis[0]].abs().sum()) == 0.0
    assert ops._lib.async_errors() == 0


@pytest.mark.parametrize("kind,cam,crops", [
    ("lego", 7, [(320, 336, 160, 96), (96, 400, 128, 64)]),
    ("garden", 0, [(880, 560, 160, 96), (48, 640, 128, 64), (1776, 464, 144, 80)]),
    ("6m", 3, [(896, 592, 128, 64)])])
def test_oracle_on_camera_crops_of_the_full_scene(dev, kind, cam, crops):
    """The float64 oracle can not run a 2 M-Gaussian 1080p frame in test time, but it can run WINDOWS of that
    frame: the same Gaussians, the same camera with its principal point shifted.  The HIP path runs the
    identical cropped camera over all N Gaussians (projection, binning, rasteriser, both backward passes)."""
    sc = _scene(kind)
    A = activated(sc.params, torch.float32)
    for (x0, y0, cw, ch) in crops:
        Kc = crop_camera(sc.Ks[cam:cam + 1], x0, y0)
        vm = sc.viewmats[cam:cam + 1]
        with torch.no_grad():       # the oracle's own cull decides which Gaussians its autograd pass has to carry
            proj64 = O.projection(A["means"].double(), A["quats"].double(), A["scales"].double(), vm.double(), Kc.double(),
                                  cw, ch, opacities=A["opacities"].double())
            radii = proj64[0]
        idx = torch.nonzero((radii > 0).all(-1)[0]).flatten()
        assert idx.numel() > 500
        g = torch.Generator().manual_seed(x0 + y0)
        wr = torch.randn(1, ch, cw, 3, generator=g, dtype=torch.float64)
        wa = torch.randn(1, ch, cw, 1, generator=g, dtype=torch.float64)
        bg = torch.rand(1, 3, generator=g, dtype=torch.float64)
        # first pass of the library: the float32 records its lists were built from (centres, radii, depths)
        _, _, _, meta = _hip_fwd_bwd(A, vm, Kc, cw, ch, wr, wa, dev, bg=bg)
        sp32 = meta["splats"][0].cpu()[idx]
        rad32 = meta["radii"][0].cpu()[idx]
        vis64 = (radii[0][idx] > 0).all(-1)
        assert int(((rad32 > 0).all(-1) != vis64).sum()) == 0
        r_ref, a_ref, g_ref, mask = _oracle_fwd_bwd({k: v[idx] for k, v in A.items()}, vm, Kc, cw, ch, wr, wa, bg=bg,
                                                    lists_from=(sp32[None, :, 0:2], rad32[None], sp32[None, :, 9]), mask_marginal=True)
        n_masked = int(mask.sum())
        assert n_masked <= 0.002 * cw * ch, (kind, (x0, y0), "pixels with a float32-unresolvable 1/255 decision:", n_masked)
        keep = (~mask)[..., None].double()
        # second pass: the same loss as the oracle's -- every pixel whose decisions float32 resolves, every Gaussian held to 2e-3
        r, a, gr, meta = _hip_fwd_bwd(A, vm, Kc, cw, ch, wr * keep, wa * keep, dev, bg=bg)
        _check_images(r * keep.float(), a * keep.float(), r_ref * keep, a_ref * keep, mean_tol=1e-4, q_tol=2e-3, max_tol=5e-3)
        assert a_ref.mean() > 0.05
        for k in ("means", "quats", "scales", "opacities", "sh"):
            e_all = rel_err(gr[k][idx], g_ref[k])
            assert e_all < 2e-3, (kind, (x0, y0), k, e_all, "masked pixels:", n_masked)
            rest = gr[k].clone()
            rest[idx] = 0
            assert float(rest.norm()) <= 1e-3 * float(gr[k].norm()), (k, "gradient outside the oracle's visible set")
        wr, wa = wr * keep, wa * keep
        # the rasteriser alone: the oracle's compositing and autograd on the HIP projection's own records
        sp = meta["splats"][0].cpu()[idx].double()
        leaves = [sp[None, :, a:b].clone().requires_grad_(True) for a, b in ((0, 2), (2, 5), (6, 9))]
        op = sp[None, :, 5].clone().requires_grad_(True)
        tw, th = math.ceil(cw / 16), math.ceil(ch / 16)
        _, ids2, flat2 = O.isect_tiles(leaves[0], meta["radii"][0].cpu()[idx][None], sp[None, :, 9], 16, tw, th)
        r2, a2, _ = O.rasterize_to_pixels(leaves[0], leaves[1], leaves[2], op, cw, ch, 16,
                                          O.isect_offset_encode(ids2, 1, tw, th), flat2, backgrounds=bg)
        ((r2 * wr).sum() + (a2 * wa).sum()).backward()
        vs = meta["v_splats"][0].cpu()[idx].double()
        assert ((r.double() - r2.detach()) * keep).abs().max() < 2e-3
        for nm, got, ref in (("mean2d", vs[:, 0:2], leaves[0].grad[0]), ("conic", vs[:, 2:5], leaves[1].grad[0]),
                             ("opacity", vs[:, 5], op.grad[0]), ("colour", vs[:, 6:9], leaves[2].grad[0])):
            assert rel_err(got, ref) < 1e-3, (kind, (x0, y0), "rasteriser only", nm, rel_err(got, ref))
    assert _ops()._lib.async_errors() == 0


def test_adam_two_million_matches_torch_optim(dev):
    ops = _ops()
    from mi3dgs.trainer import WIDTHS
    N = 2_000_000
    gen = torch.Generator(device=dev).manual_seed(3)
    lrs = (1.6e-4, 1e-3, 5e-3, 5e-2, 2.5e-3, 1.25e-4)
    p = [torch.randn(N, w, device=dev, generator=gen) for w in WIDTHS]
    ref = [t.clone().requires_grad_(True) for t in p]
    opt = torch.optim.Adam([dict(params=[t], lr=lr) for t, lr in zip(ref, lrs)], eps=1e-15, betas=(0.9, 0.999))
    m = [torch.zeros_like(t) for t in p]
    v = [torch.zeros_like(t) for t in p]
    for step in range(1, 4):
        grads = [torch.randn(N, w, device=dev, generator=gen) * (10.0 ** float(torch.randint(-6, 1, (1,)))) for w in WIDTHS]
        grads[5][::3] = 0.0                                   # exact zeros: the eps = 1e-15 corner
        for t, gq in zip(ref, grads):
            t.grad = gq.clone()
        opt.step()
        ops.adam_step(p, grads, m, v, lrs, step, eps=1e-15)
    for t, r in zip(p, ref):
        assert rel_err(t, r.detach()) < 1e-6
        assert bool(torch.isfinite(t).all())


@pytest.mark.parametrize("kind,cam,absgrad,with_bg", [("lego", 3, True, True), ("garden", 1, False, False), ("garden", 2, True, True)])
def test_the_two_backward_rasterisers_agree_at_full_size(dev, kind, cam, absgrad, with_bg):
    """The product backward contracts its per-splat pixel sums on the matrix pipe, carrying every term as two bf16 values
    (csrc/rasterize_bwd_mm.hip); the reduce-scatter kernel of the first half of round 2 sums the same terms in float32 on
    the vector pipe (experiments library, mode 3).  Same forward, same lists: the gradient records must agree to the transport's
    2^-16 per term -- checked on the full S1 / S2 scenes, with and without background and |d/dxy| sums."""
    ops = _ops()
    sc = _scene(kind)
    W, H = sc.width, sc.height
    g, vm, K, radii, splats, keys = _project(sc, dev, cam, want_keys=True)
    b = ops.bin_tiles(radii, splats, W, H, 16, tight=True)
    bg = torch.tensor([[0.3, 0.6, 0.1]], device=dev) if with_bg else None
    r, a, l = ops.rasterize_fwd(splats, b, W, H, 16, bg, {})
    gen = torch.Generator().manual_seed(9)
    vr = (torch.rand(1, H, W, 3, generator=gen) - 0.5).to(dev)
    va = (torch.rand(1, H, W, 1, generator=gen) - 0.5).to(dev)
    # the product backward from the product library; the all-f32 reduce-scatter backward only exists in the experiments
    # build (libmi3dgs_exp.so, mode 3), called here through its own handle on the same device buffers
    import ctypes as C
    mm_out = ops.rasterize_bwd(splats, b, W, H, a, l, vr, va, 16, bg, absgrad).clone()
    ex = ops._lib.experiments_lib()
    assert ex.mi3dgs_debug_set_raster_mode(3) == 0
    rs_out = torch.zeros_like(mm_out)
    try:
        ops._lib.exp_call("mi3dgs_rasterize_bwd", 1, W, H, 16, b["tile_width"], b["tile_height"], ops._p(splats), ops._p(b["isect_offsets"]),
                          ops._p(b["flatten_ids"]), ops._p(b["n_isect"]), ops._p(bg), ops._p(a), ops._p(l), ops._p(vr), ops._p(va),
                          int(absgrad), ops._p(rs_out), int(splats.shape[1]), None, None, 0, ops._stream(dev))
        torch.cuda.synchronize()
    finally:
        ex.mi3dgs_debug_set_raster_mode(1)
    outs = [mm_out, rs_out]
    mm, rs = outs
    assert bool(torch.isfinite(mm).all()) and float(rs.abs().sum()) > 0
    ncol = 11 if absgrad else 9
    for c in range(ncol):
        x, y = mm[0, :, c].double(), rs[0, :, c].double()
        assert float((x - y).norm() / y.norm().clamp(min=1e-30)) < 2e-5, (c, float((x - y).norm() / y.norm()))
    # per record: the difference stays at the level of the transport (and of float atomics landing in another order)
    d = (mm[0, :, :ncol] - rs[0, :, :ncol]).abs().double()
    scale = rs[0, :, :ncol].abs().double().amax(dim=0, keepdim=True).clamp(min=1e-30)
    assert float((d / scale).max()) < 1e-4
    assert ops._lib.async_errors() == 0


@pytest.mark.parametrize("kind,cam,absgrad", [("lego", 3, False), ("lego", 5, True)])
def test_backward_in_segments_agrees_with_the_serial_walk_at_full_size(dev, kind, cam, absgrad):
    """S1, full frame: the forward with the segment workspace leaves hundreds of checkpoints, the backward walks the segments as work
    items of their own (WIDE shape) -- same gradient records as one block per tile, up to the float32 rounding of
    (final colour - checkpoint colour) and the order of the float atomics; and the counter is clear again afterwards."""
    ops = _ops()
    sc = _scene(kind)
    W, H = sc.width, sc.height
    g, vm, K, radii, splats, keys = _project(sc, dev, cam, want_keys=True)
    b = ops.bin_tiles(radii, splats, W, H, 16, tight=True)
    bg = torch.tensor([[0.3, 0.6, 0.1]], device=dev)
    gen = torch.Generator().manual_seed(10)
    vr = (torch.rand(1, H, W, 3, generator=gen) - 0.5).to(dev)
    va = (torch.rand(1, H, W, 1, generator=gen) - 0.5).to(dev)
    r0, a0, l0 = [t.clone() for t in ops.rasterize_fwd(splats, b, W, H, 16, bg, {})]
    serial = ops.rasterize_bwd(splats, b, W, H, a0, l0, vr, va, 16, bg, absgrad).clone()
    ws = ops.raster_seg_workspace(b, 1, dev)
    r1, a1, l1 = ops.rasterize_fwd(splats, b, W, H, 16, bg, {}, seg_ws=ws)
    assert torch.equal(r0, r1) and torch.equal(a0, a1) and torch.equal(l0, l1)          # the checkpoints change nothing in the forward
    n_items = seg_ctl(ws)["items"]
    assert n_items > 100, n_items
    seg = ops.rasterize_bwd(splats, b, W, H, a1, l1, vr, va, 16, bg, absgrad, render=r1, seg_ws=ws)
    assert_seg_clear(ws)
    ncol = 11 if absgrad else 9
    assert bool(torch.isfinite(seg).all())
    for c in range(ncol):
        x, y = seg[0, :, c].double(), serial[0, :, c].double()
        assert float((x - y).norm() / y.norm().clamp(min=1e-30)) < 3e-4, (c, float((x - y).norm() / y.norm()))
    assert ops._lib.async_errors() == 0


def _wolf_frame(dev, opacity_scale):
    """The real-training regime of docs/FINDINGS_r03.md 4.2: the reference's wolf.spz inside an opaque shell, 960 x 720 (tools/train_wolf.py)."""
    import math
    from helpers import load_wolf
    from mi3dgs import scenes
    P = load_wolf()
    centre = P["means"].median(0).values
    ext = float((P["means"] - centre).abs().quantile(0.99))
    W, H = 960, 720
    P = scenes.add_backdrop(scenes.Scene("wolf", P, None, None, W, H), 12000, 9.0 * ext, tuple(centre.tolist())).params
    eye = centre + torch.tensor([3.2 * ext * math.cos(0.6) * math.cos(0.3), -3.2 * ext * math.sin(0.3), 3.2 * ext * math.sin(0.6) * math.cos(0.3)])
    if opacity_scale != 1.0:
        P["opacities"] = torch.logit((torch.sigmoid(P["opacities"]) * opacity_scale).clamp(1e-4, 0.999))
    return scenes.Scene("wolf", P, scenes.look_at(eye, centre, up=(0.0, -1.0, 0.0))[None], scenes._intrinsics(1.25 * W, W, H)[None], W, H)


@pytest.mark.parametrize("kind,cam,opacity_scale", [("lego", 3, 1.0), ("wolf", 0, 1.0), ("wolf", 0, 0.1)])
def test_forward_in_segments_agrees_with_the_serial_forward_at_full_size(dev, kind, cam, opacity_scale):
    """The opt-in forward in segments on whole frames: S1 (lists of hundreds), the wolf frame as it is (pixels saturate early:
    most heavy tiles have pixels stopping in several segments) and with every opacity at a tenth (lists walked to their ends).
    Render, alpha and last contributor against the serial forward; then the backward from what the segmented forward left
    (checkpoints written by the combine pass, items of segments nobody walked past switched off) against the serial backward."""
    ops = _ops()
    sc = _scene(kind) if kind != "wolf" else _wolf_frame(dev, opacity_scale)
    W, H = sc.width, sc.height
    g, vm, K, radii, splats, keys = _project(sc, dev, cam)
    b = ops.bin_tiles(radii, splats, W, H, 16, tight=True)
    bg = torch.tensor([[0.3, 0.6, 0.1]], device=dev)
    gen = torch.Generator().manual_seed(10)
    vr = (torch.rand(1, H, W, 3, generator=gen) - 0.5).to(dev)
    va = (torch.rand(1, H, W, 1, generator=gen) - 0.5).to(dev)
    r0, a0, l0 = [t.clone() for t in ops.rasterize_fwd(splats, b, W, H, 16, bg, {})]
    serial = ops.rasterize_bwd(splats, b, W, H, a0, l0, vr, va, 16, bg, False).clone()
    ws = ops.raster_seg_workspace(b, 1, dev)
    try:
        ops.set_raster_fwd_segments(True)
        for rep in range(2):             # twice: a forward with no backward behind it must leave nothing in the next one's way
            r1, a1, l1 = [t.clone() for t in ops.rasterize_fwd(splats, b, W, H, 16, bg, {}, seg_ws=ws)]
        ctl = seg_ctl(ws)
        seg = ops.rasterize_bwd(splats, b, W, H, a1, l1, vr, va, 16, bg, False, render=r1, seg_ws=ws)
    finally:
        ops.set_raster_fwd_segments(False)
    off = b["isect_offsets"].flatten().cpu().long()
    lens = torch.cat([off[1:], b["n_isect"].cpu().long().reshape(1)]) - off
    heavy = lens > 256
    assert ctl["heavy"] == int(heavy.sum()) > 50 and ctl["items"] == int(((lens[heavy] + 255) // 256).sum()), (ctl, int(heavy.sum()))
    assert_seg_clear(ws)
    assert float((r1 - r0).abs().max()) < 2e-4 and float((a1 - a0).abs().max()) < 2e-4
    assert int(((r1 - r0).abs().amax(-1) > 3e-6).sum()) <= 8 and int((l1 != l0).sum()) <= 8, \
        (int(((r1 - r0).abs().amax(-1) > 3e-6).sum()), int((l1 != l0).sum()))
    assert bool(torch.isfinite(seg).all())
    for c in range(9):
        x, y = seg[0, :, c].double(), serial[0, :, c].double()
        assert float((x - y).norm() / y.norm().clamp(min=1e-30)) < 3e-4, (c, float((x - y).norm() / y.norm()))
    assert ops._lib.async_errors() == 0


def test_s2_training_step_properties_and_fused_adam(dev):
    """One whole training step of the bench configuration (2 M Gaussians, 1080p, capacity mode, fused
    binning, fused backward + Adam) against the unfused sequence on the same inputs; then the properties."""
    ops = _ops()
    from mi3dgs import trainer
    from mi3dgs.trainer import GROUPS
    sc = _scene("garden")
    g = sc.to(dev)
    n = g.params["means"].shape[0]
    vm, ks = g.viewmats[:2].contiguous(), g.Ks[:2].contiguous()
    tgt = torch.rand(2, sc.height, sc.width, 3, device=dev)
    cfgs = [trainer.TrainConfig(capacity=n, refine_start_iter=10 ** 9, max_isect=40_000_000, fuse_adam=f, refine_scale2d_stop_iter=4000)
            for f in (True, False)]            # (step 3001 < 4000: the screen-radius statistic is written too)
    trs = [trainer.Trainer(g.params, vm, ks, tgt, sc.width, sc.height, c) for c in cfgs]
    for tr in trs:
        tr.step_count = 3001
        loss = tr.step(1, want_loss=True)
        assert math.isfinite(loss) and 0.0 < loss < 2.0
        I = int(tr.last["binning"]["n_isect"].item())
        assert 5_000_000 < I < 40_000_000
    for gname in GROUPS:
        a, b = trs[0].model.p(gname), trs[1].model.p(gname)
        assert bool(torch.isfinite(a).all()) and rel_err(a, b) < 1e-5
        assert float((a - g.params[gname].view_as(a)).abs().max()) > 0.0            # the step moved them
        # the first moments are 0.1 x the gradient, the second 0.001 x its square: the fused kernel's
        # gradients (never written to memory) against the unfused pair's, up to float-atomic order
        for k in ("m", "v"):
            assert rel_err(trs[0].model.state(gname, k), trs[1].model.state(gname, k)) < 2e-3, (gname, k)
    for k in ("grad2d", "count", "radii"):
        assert rel_err(trs[0].stats[k], trs[1].stats[k]) < 1e-4
    al = trs[0].raster_out["alphas"]
    assert float(al.min()) >= 0.0 and float(al.max()) <= 1.0
    trs[0].check_async_errors()


def test_isect_capacity_overflow_is_clamped_and_reported(dev):
    """max_isect deliberately too small: nothing may read past the buffers (the published count is the
    capacity), the render still runs, and the sticky device word says so."""
    ops = _ops()
    sc = _scene("lego")
    g, vm, K, radii, splats, keys = _project(sc, dev, 3, want_keys=True)
    W, H = sc.width, sc.height
    full = ops.bin_tiles(radii, splats, W, H, 16, tight=True)
    I = int(full["n_isect"].item())
    cap = I // 3
    for fused in (True, False):
        b = ops.bin_tiles(radii, splats, W, H, 16, max_isect=cap, tight=True, fused=fused,
                          depth_keys=keys.clone() if fused else None)
        assert int(b["n_isect"].item()) == cap
        assert int(b["isect_offsets"].max()) <= cap
        r, a, l = ops.rasterize_fwd(splats, b, W, H, 16, None, {})
        assert bool(torch.isfinite(r).all()) and int(l.max()) < cap
        v = ops.rasterize_bwd(splats, b, W, H, a, l, torch.ones_like(r), torch.zeros_like(a), 16, None)
        assert bool(torch.isfinite(v).all())
        assert ops._lib.async_errors() & 4
    assert ops._lib.async_errors() == 0          # reading cleared it


# ------------------------------------------------------------------------- committed fixtures
def test_golden_tiny_scene_on_the_gpu(dev):
    """tests/golden/tiny_scene.pt (oracle/make_golden.py): the HIP path against the committed vectors."""
    import os
    import mi3dgs
    from mi3dgs import ops
    G = torch.load(os.path.join(os.path.dirname(__file__), "golden", "tiny_scene.pt"), weights_only=False)
    A = activated(G["inputs"], torch.float32)
    gl = {k: v.to(dev).requires_grad_(True) for k, v in A.items()}
    r, a, meta = mi3dgs.rasterization(gl["means"], gl["quats"], gl["scales"], gl["opacities"], gl["sh"],
                                      G["viewmats"].to(dev), G["Ks"].to(dev), G["width"], G["height"], sh_degree=3,
                                      backgrounds=G["backgrounds"].to(dev), want_isect_ids=True)
    assert (r.detach().cpu() - G["render"]).abs().max() < 2e-3 and (a.detach().cpu() - G["alphas"]).abs().max() < 2e-3
    assert torch.equal(meta["radii"].cpu(), G["radii"])
    assert torch.equal(meta["flatten_ids"].cpu(), G["flatten_ids"]) and torch.equal(meta["isect_offsets"].cpu(), G["isect_offsets"])
    assert torch.equal(meta["isect_ids"].cpu() >> 32, G["isect_ids"] >> 32)
    sums, scratch = ops.loss_fwd(r.detach().contiguous(), G["target"].to(dev))
    n = r.numel()
    assert abs(float(ops.loss_value(sums, n, 0.2)) - G["loss"]) < 1e-5
    v = ops.loss_bwd(r.detach().contiguous(), G["target"].to(dev), scratch, 0.2, 1.0)
    r.backward(v)
    for k in ("means", "quats", "scales", "opacities", "sh"):
        assert rel_err(gl[k].grad.cpu(), G["grads"][k]) < 2e-3, k


def test_reference_held_wolf_matches_oracle(dev):
    """A REAL trained splat (the reference's source/Gradio/favorites/wolf.spz, decoded by the reference's own
    codec): thin anisotropic Gaussians, real SH, opacities up to 254/255.  Forward and gradients against the
    float64 oracle on three small cameras."""
    sc = wolf_scene(n_views=3, width=96, height=64)
    assert sc.params["means"].shape[0] == 90_586
    A = activated(sc.params, torch.float32)
    g = torch.Generator().manual_seed(21)
    wr = torch.randn(3, 64, 96, 3, generator=g, dtype=torch.float64)
    wa = torch.randn(3, 64, 96, 1, generator=g, dtype=torch.float64)
    bg = torch.rand(3, 3, generator=g, dtype=torch.float64)
    r_ref, a_ref, g_ref = _oracle_fwd_bwd(A, sc.viewmats, sc.Ks, 96, 64, wr, wa, bg=bg)
    r, a, gr, _ = _hip_fwd_bwd(A, sc.viewmats, sc.Ks, 96, 64, wr, wa, dev, bg=bg)
    _check_images(r, a, r_ref, a_ref, mean_tol=1e-4, q_tol=5e-3, max_tol=5e-2)
    assert a_ref.max() > 0.95 and 0.05 < a_ref.mean() < 0.9
    for k in ("means", "quats", "scales", "opacities", "sh"):
        e = rel_err(gr[k], g_ref[k])
        assert e < 3e-3, (k, e)


def test_wolf_full_frame_properties(dev):
    """The same real splat at 1080p: integer work bit-exact against the torch restatement, tight = box render."""
    ops = _ops()
    sc = wolf_scene(n_views=2, width=1920, height=1080)
    W, H = 1920, 1080
    tw, th = 120, 68
    g, vm, K, radii, splats, keys = _project(sc, dev, 1, want_keys=True)
    tpg_r, ids_r, flat_r, offs_r = isect_reference(radii, splats, 16, tw, th)
    I = ids_r.numel()
    b = ops.bin_tiles(radii, splats, W, H, 16, want_isect_ids=True, want_tiles_per_gauss=True, tight=False)
    assert int(b["n_isect"].item()) == I > 100_000
    assert torch.equal(b["flatten_ids"], flat_r) and torch.equal(b["isect_ids"], ids_r) and torch.equal(b["isect_offsets"], offs_r)
    bt = ops.bin_tiles(radii, splats, W, H, 16, max_isect=I, tight=True, fused=True, depth_keys=keys.clone())
    r_b, a_b, _ = ops.rasterize_fwd(splats, b, W, H, 16, None, {})
    r_t, a_t, _ = ops.rasterize_fwd(splats, bt, W, H, 16, None, {})
    assert torch.equal(r_b, r_t) and torch.equal(a_b, a_t)
    assert float(a_t.max()) > 0.99 and ops._lib.async_errors() == 0


# --------------------------------------------------------- the alpha >= 1/255 decision, fwd vs bwd
def test_forward_and_backward_agree_on_threshold_splats(dev):
    """Splats built to land within a few ulp of alpha = 1/255 at one pixel each.  The forward rasteriser
    decides membership once; the backward replays the list and must take the SAME decision, or its
    transmittance replay T /= (1 - alpha) drifts.  Observable without any debug hook:
    splat j is white, sits alone on pixel j behind nothing, so render(j) > 0 iff the forward composited it,
    and its colour gradient is non-zero iff the backward did."""
    ops = _ops()
    tw, th = 12, 10
    W, H = tw * 16, th * 16
    n = W * H
    gen = torch.Generator().manual_seed(77)
    px = torch.arange(W).repeat(H).float() + 0.5
    py = torch.arange(H).repeat_interleave(W).float() + 0.5
    # centre within 0.2 px of its pixel and a mild cross term: every OTHER pixel then sees sigma at least 0.1
    # larger, i.e. alpha 10 % under the threshold, so each splat can only ever touch its own pixel
    off = torch.rand(n, 2, generator=gen) * 0.4 - 0.2
    ca = 0.5 + 3.0 * torch.rand(n, generator=gen)
    cc = 0.5 + 3.0 * torch.rand(n, generator=gen)
    cb = (torch.rand(n, generator=gen) - 0.5) * 0.4 * torch.sqrt(ca * cc)
    dx, dy = off[:, 0].double(), off[:, 1].double()
    sigma = 0.5 * (ca.double() * dx * dx + cc.double() * dy * dy) + cb.double() * dx * dy
    opac = ((1.0 / 255.0) * torch.exp(sigma)).float()
    ulps = torch.randint(-3, 4, (n,), generator=gen)
    opac = (opac.view(torch.int32) + ulps.int()).view(torch.float32)       # a few ulp either side of the threshold
    S = torch.zeros(1, n, ops.SPLAT_STRIDE)
    S[0, :, 0], S[0, :, 1] = px + off[:, 0], py + off[:, 1]
    S[0, :, 2], S[0, :, 3], S[0, :, 4], S[0, :, 5] = ca, cb, cc, opac
    S[0, :, 6:9] = 1.0
    S[0, :, 9] = 1.0 + torch.rand(n, generator=gen)
    radii = torch.ones(1, n, 2, dtype=torch.int32)                         # one-pixel footprint: only its own tile
    S, radii = S.to(dev), radii.to(dev)
    b = ops.bin_tiles(radii, S, W, H, 16, tight=False)
    r, a, l = ops.rasterize_fwd(S, b, W, H, 16, None, {})
    v = ops.rasterize_bwd(S, b, W, H, a, l, torch.ones_like(r), torch.zeros_like(a), 16, None)
    fwd_in = r[0, ..., 0].flatten() > 0
    bwd_in = v[0, :, 6] != 0
    assert 0.2 < float(fwd_in.float().mean()) < 0.8                        # the construction straddles the threshold
    assert torch.equal(fwd_in, bwd_in), f"{int((fwd_in != bwd_in).sum())} of {n} threshold splats decided differently"
    # where both composited it, the backward's alpha is the forward's: colour gradient = alpha * T = render.  The backward carries
    # alpha * T to its pixel sums as two bf16 terms (16 significant bits, csrc/rasterize_bwd_mm.hip): 2^-16 = 1.5e-5 relative.
    both = fwd_in & bwd_in
    assert torch.allclose(v[0, :, 6][both], r[0, ..., 0].flatten()[both], rtol=2e-5, atol=0)
