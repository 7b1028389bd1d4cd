"""GPU parity: every stage of the HIP path, called through the C-ABI, against the CPU oracle."""
import math

import numpy as np
import pytest
import torch

from helpers import activated, assert_clean, assert_seg_clear, rel_err, seg_ctl, small_scene
from oracle import gs_oracle as O

pytestmark = pytest.mark.gpu


def _ops():
    from mi3dgs import ops
    return ops


# ----------------------------------------------------------------- scan / sort (bit exact)
@pytest.mark.parametrize("n", [1, 63, 4096, 4097, 100_003, 2_000_001])
def test_scan_exclusive_bit_exact(dev, n):
    ops = _ops()
    x = torch.randint(0, 50, (n,), dtype=torch.int32)
    out = ops.scan_exclusive_u32(x.to(dev))
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.scan_exclusive_u32(x.to(dev), total=total)
    ref = np.concatenate([[0], np.cumsum(x.numpy().astype(np.int64))[:-1]])
    assert np.array_equal(out.cpu().numpy().astype(np.int64), ref)
    assert int(total.item()) == int(x.sum())


@pytest.mark.parametrize("mode", [0, 1, 3])       # classic multi-kernel passes / onesweep (all passes in one launch up to 96 tiles) / onesweep, one launch per pass
@pytest.mark.parametrize("n,nbits", [(1, 32), (64, 32), (4096, 32), (5000, 13), (100_003, 32), (196_608, 32), (196_609, 32),
                                     (600_000, 16), (900_000, 32), (1_500_000, 13), (1_600_000, 13), (300_000, 7)])
def test_radix_sort_stable_bit_exact(dev, n, nbits, mode):
    ops = _ops()
    ops._lib.lib().mi3dgs_debug_set_sort_mode(mode)
    g = torch.Generator().manual_seed(n)
    hi = (1 << nbits) - 1
    # many duplicates so that stability is exercised
    keys = torch.randint(0, min(hi, 5000) + 1, (n,), generator=g, dtype=torch.int64)
    if nbits == 32:
        keys = keys * 858_993            # spread over the high bits
    vals = torch.arange(n, dtype=torch.int32)
    k = keys.to(torch.int32).to(dev) if nbits < 32 else (keys & 0xFFFFFFFF).to(torch.int64).to(torch.int32).to(dev)
    v = vals.to(dev)
    ops.sort_pairs_u32(k, v, nbits)
    ops._lib.lib().mi3dgs_debug_set_sort_mode(2)
    assert_clean(ops, f"sort n={n} nbits={nbits} mode={mode}")     # a barrier / look-back wait that ran out only shows in the error word
    order = np.argsort(keys.numpy(), kind="stable")
    assert np.array_equal(v.cpu().numpy(), vals.numpy()[order])
    assert np.array_equal(k.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, keys.numpy()[order])


# ------------------------------------------------------------------------- projection + SH
@pytest.mark.parametrize("sh_degree", [0, 1, 2, 3])
def test_project_fwd_matches_oracle(dev, sh_degree):
    ops = _ops()
    sc = small_scene(n=600, seed=3)
    sc.params["means"][:40] *= 6.0          # some behind the cameras / outside the frustum
    sc.params["opacities"][40:60] = -7.0    # some below the 1/255 opacity cut
    A = activated(sc.params)
    radii_r, m2d_r, dep_r, con_r, _ = O.projection(A["means"], A["quats"], A["scales"], sc.viewmats.double(),
                                                   sc.Ks.double(), sc.width, sc.height, opacities=A["opacities"])
    g = sc.to(dev)
    radii, splats = ops.project_fwd(g.params["means"], g.params["quats"], g.params["scales"], g.params["opacities"],
                                    g.viewmats, g.Ks, sc.width, sc.height, sh0=g.params["sh0"], shN=g.params["shN"],
                                    sh_degree=sh_degree, flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC)
    radii, splats = radii.cpu(), splats.cpu()
    vis_r = (radii_r > 0).all(-1)
    vis = (radii > 0).all(-1)
    # cull decisions and integer radii: allow rounding flips on a vanishing fraction only
    assert (vis != vis_r).float().mean() < 2e-3
    both = vis & vis_r
    assert both.sum() > 200
    assert ((radii != radii_r).any(-1) & both).float().mean() < 5e-3
    assert torch.allclose(splats[..., 0:2][both].double(), m2d_r[both], rtol=1e-4, atol=2e-3)
    assert torch.allclose(splats[..., 9][both].double(), dep_r[both], rtol=1e-5, atol=1e-5)
    assert rel_err(splats[..., 2:5][both], con_r[both]) < 1e-4
    assert torch.allclose(splats[..., 5][both].double(), A["opacities"][None].expand_as(vis)[both], rtol=1e-5, atol=1e-6)
    campos = torch.linalg.inv(sc.viewmats.double())[:, :3, 3]
    dirs = A["means"][None] - campos[:, None]
    cols = torch.clamp(O.spherical_harmonics(sh_degree, dirs, A["sh"][None].expand(len(campos), -1, -1, -1)) + 0.5, min=0)
    assert torch.allclose(splats[..., 6:9][both].double(), cols[both], rtol=1e-4, atol=1e-5)
    # culled records are zero
    assert float(splats[~vis].abs().sum()) == 0


def test_project_fwd_plain_colors_and_activated_inputs(dev):
    ops = _ops()
    sc = small_scene(n=300, seed=4)
    A = activated(sc.params, torch.float32)
    cols = torch.rand(300, 3)
    radii, splats = ops.project_fwd(A["means"].to(dev), A["quats"].to(dev), A["scales"].to(dev).contiguous(),
                                    A["opacities"].to(dev).contiguous(), sc.viewmats.to(dev), sc.Ks.to(dev), sc.width,
                                    sc.height, colors=cols.to(dev))
    vis = (radii > 0).all(-1).cpu()
    assert torch.equal(splats[..., 6:9].cpu()[vis], cols[None].expand(2, -1, -1)[vis])


# ------------------------------------------------------------------------- tile binning
@pytest.mark.parametrize("big,n_views", [(False, 1), (True, 2), (True, 3)])
def test_binning_bit_exact_vs_oracle(dev, big, n_views):
    ops = _ops()
    sc = small_scene(n=700, seed=5, big=big, n_views=n_views, width=80, height=56)
    g = sc.to(dev)
    radii, splats = ops.project_fwd(g.params["means"], g.params["quats"], g.params["scales"], g.params["opacities"],
                                    g.viewmats, g.Ks, sc.width, sc.height, sh0=g.params["sh0"], shN=g.params["shN"],
                                    sh_degree=3, flags=3)
    b = ops.bin_tiles(radii, splats, sc.width, sc.height, 16, want_isect_ids=True, want_tiles_per_gauss=True)
    # oracle binning on the HIP stage's own outputs => integer work must match bit for bit
    sp = splats.cpu()
    tpg, ids, flat = O.isect_tiles(sp[..., 0:2], radii.cpu(), sp[..., 9], 16, b["tile_width"], b["tile_height"])
    offs = O.isect_offset_encode(ids, n_views, b["tile_width"], b["tile_height"])
    assert int(b["n_isect"].item()) == ids.numel() > 0
    assert torch.equal(b["tiles_per_gauss"].cpu(), tpg)
    assert torch.equal(b["flatten_ids"].cpu(), flat)
    assert torch.equal(b["isect_ids"].cpu(), ids)
    assert torch.equal(b["isect_offsets"].cpu(), offs)
    # capacity mode (no host sync) gives the same live prefix
    cap = ids.numel() + 1000
    b2 = ops.bin_tiles(radii, splats, sc.width, sc.height, 16, max_isect=cap)
    assert torch.equal(b2["flatten_ids"][: ids.numel()].cpu(), flat)
    assert torch.equal(b2["isect_offsets"].cpu(), offs)


@pytest.mark.parametrize("tight", [False, True])
@pytest.mark.parametrize("n,n_views,big", [(700, 1, False), (3000, 2, True), (5, 1, True), (70000, 1, False)])
def test_fused_binning_equals_the_two_phase_path(dev, tight, n, n_views, big):
    """mi3dgs_bin_tiles (depth sort + chained count/emit) against mi3dgs_bin_count + mi3dgs_bin_emit, bit for bit."""
    ops = _ops()
    sc = small_scene(n=n, seed=17, big=big, n_views=n_views, width=200 if n > 10000 else 80, height=120 if n > 10000 else 56)
    g = sc.to(dev)
    radii, splats = ops.project_fwd(g.params["means"], g.params["quats"], g.params["scales"], g.params["opacities"],
                                    g.viewmats, g.Ks, sc.width, sc.height, sh0=g.params["sh0"], shN=g.params["shN"],
                                    sh_degree=3, flags=3)
    ref = ops.bin_tiles(radii, splats, sc.width, sc.height, 16, want_isect_ids=True, want_tiles_per_gauss=True, tight=tight)
    I = int(ref["n_isect"].item())
    assert I > 0
    # sort keys straight from the projection (consumed by the fused call, so a fresh copy each time)
    keys = torch.empty(radii.shape[:2], dtype=torch.int32, device=dev)
    r2, s2 = ops.project_fwd(g.params["means"], g.params["quats"], g.params["scales"], g.params["opacities"],
                             g.viewmats, g.Ks, sc.width, sc.height, sh0=g.params["sh0"], shN=g.params["shN"],
                             sh_degree=3, flags=3, depth_keys=keys)
    assert torch.equal(r2, radii) and torch.equal(s2, splats)
    vis = (radii > 0).all(-1)
    assert torch.equal(keys[vis], splats[..., 9][vis].view(torch.int32)) and bool((keys[~vis] == -1).all())
    for cap in (I + 777, I, max(I // 2, 1)):                 # roomy, exact, overflowing capacity
        for fused in (True, False, "keys"):
            b = ops.bin_tiles(radii, splats, sc.width, sc.height, 16, max_isect=cap, want_isect_ids=True,
                              want_tiles_per_gauss=True, tight=tight, fused=bool(fused),
                              depth_keys=keys.clone() if fused == "keys" else None)
            # the published count is clamped to the capacity (no later kernel may walk past the buffers);
            # an overflow is reported through the sticky device word, bit 4
            assert int(b["n_isect"].item()) == min(I, cap)
            assert (ops._lib.async_errors() & 4) == (4 if cap < I else 0)
            assert torch.equal(b["tiles_per_gauss"], ref["tiles_per_gauss"])
            if cap >= I:
                assert torch.equal(b["flatten_ids"][:I], ref["flatten_ids"])
                assert torch.equal(b["tile_keys"][:I], ref["tile_keys"])
                assert torch.equal(b["isect_ids"][:I], ref["isect_ids"])
                assert torch.equal(b["isect_offsets"], ref["isect_offsets"])
            else:
                # overflow: the first `cap` emitted pairs (depth order) are kept, sorted by tile
                k = b["tile_keys"][:cap]
                assert bool((k[1:] >= k[:-1]).all())


def test_binning_empty_scene(dev):
    ops = _ops()
    sc = small_scene(n=50, seed=6)
    g = sc.to(dev)
    far = g.params["means"] + 1000.0          # everything behind / outside
    radii, splats = ops.project_fwd(far, g.params["quats"], g.params["scales"], g.params["opacities"], g.viewmats,
                                    g.Ks, sc.width, sc.height, sh0=g.params["sh0"], shN=g.params["shN"], sh_degree=3,
                                    flags=3)
    assert int((radii > 0).sum()) == 0
    b = ops.bin_tiles(radii, splats, sc.width, sc.height, 16)
    assert int(b["n_isect"].item()) == 0 and int(b["isect_offsets"].abs().sum()) == 0
    b = ops.bin_tiles(radii, splats, sc.width, sc.height, 16, max_isect=1000)       # fused path
    assert int(b["n_isect"].item()) == 0 and int(b["isect_offsets"].abs().sum()) == 0
    # the training step's call (wave-granular emit): the kernel itself writes the zero count, nothing clears it in front
    for _ in range(2):
        b = ops.bin_tiles(radii, splats, sc.width, sc.height, 16, max_isect=1000, tight=True, want_tile_keys=False)
        assert int(b["n_isect"].item()) == 0 and int(b["isect_offsets"].abs().sum()) == 0
    bg = torch.rand(2, 3, device=dev)
    r, a, _ = ops.rasterize_fwd(splats, b, sc.width, sc.height, 16, bg)
    assert torch.equal(r, bg[:, None, None, :].expand_as(r)) and float(a.abs().max()) == 0


# -------------------------------------------------------------------- full forward/backward
def _run_both(sc, dev, sh_degree=3, bg=True, mode="classic", absgrad=False):
    import mi3dgs
    A = activated(sc.params)
    leaves = {k: v.clone().requires_grad_(True) for k, v in A.items()}
    C = sc.viewmats.shape[0]
    bgs = torch.rand(C, 3, generator=torch.Generator().manual_seed(1)).double() if bg else None
    r_ref, a_ref, meta_ref = O.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"],
                                             leaves["sh"], sc.viewmats.double(), sc.Ks.double(), sc.width, sc.height,
                                             sh_degree=sh_degree, backgrounds=bgs, rasterize_mode=mode)
    gl = {k: v.detach().float().to(dev).requires_grad_(True) for k, v in A.items()}
    r, a, meta = mi3dgs.rasterization(gl["means"], gl["quats"], gl["scales"], gl["opacities"], gl["sh"],
                                      sc.viewmats.to(dev), sc.Ks.to(dev), sc.width, sc.height, sh_degree=sh_degree,
                                      backgrounds=None if bgs is None else bgs.float().to(dev), rasterize_mode=mode,
                                      absgrad=absgrad)
    return (r_ref, a_ref, meta_ref, leaves), (r, a, meta, gl)


@pytest.mark.parametrize("mode,bg,sh_degree", [("classic", True, 3), ("classic", False, 1), ("antialiased", True, 2)])
def test_rasterization_forward_matches_oracle(dev, mode, bg, sh_degree):
    sc = small_scene(n=500, seed=7, big=True, width=72, height=40)
    (r_ref, a_ref, meta_ref, _), (r, a, meta, _) = _run_both(sc, dev, sh_degree, bg, mode)
    dr = (r.cpu().double() - r_ref).abs()
    da = (a.cpu().double() - a_ref).abs()
    # f32 vs f64: a handful of pixels may flip a 1/255 skip or a radius ceil; bound both the
    # bulk (tight) and the worst case (one splat's worth)
    assert dr.mean() < 2e-5 and da.mean() < 2e-5, (dr.mean(), da.mean())
    assert torch.quantile(dr.flatten(), 0.999) < 5e-4
    assert dr.max() < 2e-2 and da.max() < 2e-2
    assert a_ref.max() > 0.5      # the scene is not trivially empty


@pytest.mark.parametrize("mode,bg,sh_degree,n_views,n", [
    ("classic", True, 3, 2, 350), ("antialiased", False, 3, 2, 350), ("classic", True, 0, 2, 350),
    # one camera = the training path's dedicated project_bwd kernel (LDS-staged shN slices);
    # 350 and 64 exercise the partial last wave and the exactly-full wave
    ("classic", True, 3, 1, 350), ("antialiased", True, 2, 1, 64), ("classic", False, 1, 1, 333), ("classic", True, 0, 1, 130)])
def test_rasterization_backward_matches_autograd_oracle(dev, mode, bg, sh_degree, n_views, n):
    sc = small_scene(n=n, seed=8, big=True, width=64, height=48, n_views=n_views)
    (r_ref, a_ref, _, leaves), (r, a, _, gl) = _run_both(sc, dev, sh_degree, bg, mode)
    gen = torch.Generator().manual_seed(2)
    wr = torch.randn(r_ref.shape, generator=gen).double()
    wa = torch.randn(a_ref.shape, generator=gen).double()
    ((r_ref * wr).sum() + (a_ref * wa).sum()).backward()
    ((r * wr.float().to(dev)).sum() + (a * wa.float().to(dev)).sum()).backward()
    for k in ("means", "quats", "scales", "opacities", "sh"):
        e = rel_err(gl[k].grad.cpu(), leaves[k].grad)
        assert e < 2e-3, (k, e)


@pytest.mark.parametrize("absgrad,bg,n_views", [(False, True, 1), (True, False, 2)])
def test_backward_in_segments_matches_oracle_and_the_serial_walk(dev, absgrad, bg, n_views):
    """Tiles whose lists run to thousands of entries that pixels really reach (faint splats): the forward leaves checkpoints
    at its 512-entry boundaries and the backward walks the segments as work items of their own.  Held to the autograd oracle
    like every other backward, and to the one-block-per-tile walk of the same library (segments=False)."""
    import mi3dgs
    # (a blob of 8 000 faint splats in the middle of a 192 x 160 frame: a dozen tiles with lists thousands long among 120)
    sc = small_scene(n=8000, seed=21, big=True, width=192, height=160, n_views=n_views)
    g = torch.Generator().manual_seed(5)
    sc.params["opacities"] = torch.rand(8000, generator=g) * 1.5 - 5.0          # alpha 0.7 .. 3 %: lists are walked deep
    sc.params["opacities"][:60] = 1.0                                         # and a few solid ones in between
    A = activated(sc.params)
    leaves = {k: v.clone().requires_grad_(True) for k, v in A.items()}
    bgs = torch.rand(n_views, 3, generator=g).double() if bg else None
    r_ref, a_ref, _ = O.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["sh"],
                                      sc.viewmats.double(), sc.Ks.double(), sc.width, sc.height, sh_degree=3, backgrounds=bgs)
    wr = torch.randn(r_ref.shape, generator=g).double()
    wa = torch.randn(a_ref.shape, generator=g).double()
    ((r_ref * wr).sum() + (a_ref * wa).sum()).backward()
    grads, metas = [], []
    for segments in (True, False):
        gl = {k: v.detach().float().to(dev).requires_grad_(True) for k, v in A.items()}
        r, a, meta = mi3dgs.rasterization(gl["means"], gl["quats"], gl["scales"], gl["opacities"], gl["sh"], sc.viewmats.to(dev),
                                          sc.Ks.to(dev), sc.width, sc.height, sh_degree=3,
                                          backgrounds=None if bgs is None else bgs.float().to(dev), absgrad=absgrad,
                                          segments=segments)
        if segments:          # work items the forward left; the backward's workers leave the counter clear for the next forward
            n_items = seg_ctl(meta["seg_ws"])["items"]
        ((r * wr.float().to(dev)).sum() + (a * wa.float().to(dev)).sum()).backward()
        if segments:
            assert_seg_clear(meta["seg_ws"])
        grads.append({k: gl[k].grad.cpu() for k in gl})
        metas.append(meta)
    # the case is what it claims to be: every tile's list is thousands long and hundreds of boundaries were walked past
    off = metas[0]["isect_offsets"].flatten().cpu()
    n_tiles = off.numel()
    assert int(off.diff().max()) > 3000 and n_items >= 20 * n_views and metas[1]["seg_ws"] is None, (int(off.diff().max()), n_items)
    assert n_tiles < 4096, "segments switch off for long lists on big grids only"
    for k in ("means", "quats", "scales", "opacities", "sh"):
        e_seg, e_ser = rel_err(grads[0][k], leaves[k].grad), rel_err(grads[1][k], leaves[k].grad)
        assert e_seg < 2e-3 and e_ser < 2e-3, (k, e_seg, e_ser)
        assert rel_err(grads[0][k], grads[1][k]) < 2e-4, (k, rel_err(grads[0][k], grads[1][k]))
    if absgrad:
        v0, v1 = metas[0]["v_splats"], metas[1]["v_splats"]
        assert rel_err(v0[..., 9:11].cpu(), v1[..., 9:11].cpu()) < 2e-4
        assert (v0[..., 9] * (1 + 1e-4) + 1e-9 >= v0[..., 0].abs()).all()


@pytest.mark.gpu
@pytest.mark.parametrize("bg,n_views", [(True, 2), (False, 1)])
def test_forward_in_segments_matches_oracle_and_the_serial_forward(dev, bg, n_views):
    """The opt-in forward in segments (mi3dgs_debug_set_raster_fwd_segments(1)): lists of more than 256 entries walked as
    segments side by side, strung together per pixel, the segments pixels stop in walked once more.  Same scene as above (lists
    thousands long, pixels that reach their ends, solid splats that stop others half way): render, alpha and every gradient held
    to the float64 oracle, and to the serial forward of the same library at float rounding."""
    import mi3dgs
    from mi3dgs import ops
    sc = small_scene(n=8000, seed=21, big=True, width=192, height=160, n_views=n_views)
    g = torch.Generator().manual_seed(5)
    sc.params["opacities"] = torch.rand(8000, generator=g) * 1.5 - 5.0
    sc.params["opacities"][:400] = 1.0                                        # solid ones: pixels stop in the middle of lists
    A = activated(sc.params)
    leaves = {k: v.clone().requires_grad_(True) for k, v in A.items()}
    bgs = torch.rand(n_views, 3, generator=g).double() if bg else None
    r_ref, a_ref, _ = O.rasterization(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["sh"],
                                      sc.viewmats.double(), sc.Ks.double(), sc.width, sc.height, sh_degree=3, backgrounds=bgs)
    wr = torch.randn(r_ref.shape, generator=g).double()
    wa = torch.randn(a_ref.shape, generator=g).double()
    ((r_ref * wr).sum() + (a_ref * wa).sum()).backward()
    out = []
    try:
        for fwd_segments in (True, False):
            ops.set_raster_fwd_segments(fwd_segments)
            gl = {k: v.detach().float().to(dev).requires_grad_(True) for k, v in A.items()}
            r, a, meta = mi3dgs.rasterization(gl["means"], gl["quats"], gl["scales"], gl["opacities"], gl["sh"], sc.viewmats.to(dev),
                                              sc.Ks.to(dev), sc.width, sc.height, sh_degree=3,
                                              backgrounds=None if bgs is None else bgs.float().to(dev), segments=True)
            ctl = seg_ctl(meta["seg_ws"])
            ((r * wr.float().to(dev)).sum() + (a * wa.float().to(dev)).sum()).backward()
            assert_seg_clear(meta["seg_ws"])
            out.append((r.detach().cpu(), a.detach().cpu(), meta["last_ids"].cpu(), {k: gl[k].grad.cpu() for k in gl}, ctl))
    finally:
        ops.set_raster_fwd_segments(False)
    (r1, a1, l1, g1, ctl1), (r0, a0, l0, g0, ctl0) = out
    assert ctl1["heavy"] >= 8 * n_views and ctl1["items"] >= 60 * n_views and ctl0["heavy"] == 0, (ctl1, ctl0)   # heavy tiles, segments
    assert rel_err(r1, r_ref.detach()) < 1e-4 and rel_err(a1, a_ref.detach()) < 1e-4
    # against the serial forward: float rounding; a pixel within an ulp of the 1e-4 stop may end elsewhere (bounded by 1e-4)
    assert float((r1 - r0).abs().max()) < 2e-4 and float((a1 - a0).abs().max()) < 2e-4
    assert int(((r1 - r0).abs().amax(-1) > 3e-6).sum()) <= 4 and int((l1 != l0).sum()) <= 4
    for k in ("means", "quats", "scales", "opacities", "sh"):
        e_seg = rel_err(g1[k], leaves[k].grad)
        assert e_seg < 2e-3, (k, e_seg)
        assert rel_err(g1[k], g0[k]) < 2e-4, (k, rel_err(g1[k], g0[k]))


def test_absgrad_record(dev):
    sc = small_scene(n=200, seed=9, big=True)
    (_, _, _, _), (r, a, meta, gl) = _run_both(sc, dev, 3, True, "classic", absgrad=True)
    r.sum().backward()
    v = meta["v_splats"]
    assert (v[..., 9] + 1e-12 >= v[..., 0].abs()).all() and (v[..., 10] + 1e-12 >= v[..., 1].abs()).all()
    assert float(v[..., 9].sum()) > 0


# -------------------------------------------------------------------------------- loss
@pytest.mark.parametrize("C,H,W", [(2, 45, 70), (1, 131, 301), (3, 17, 9), (1, 80, 86)])
def test_loss_fwd_bwd_matches_oracle(dev, C, H, W):
    """Sizes: fewer columns than one block owns; several column blocks and several row strips with ragged ends (131 rows in
    strips of 16, 903 floats per row in blocks of 256); an image smaller than the blur window; 258 floats per row (a second
    column block that owns two float-columns)."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    a = torch.rand(C, H, W, 3, generator=g)
    b = (a + 0.2 * torch.randn(C, H, W, 3, generator=g)).clamp(0, 1)
    ad = a.double().requires_grad_(True)
    L = O.photometric_loss(ad, b.double(), 0.2)
    L.backward()
    sums, scratch = ops.loss_fwd(a.to(dev), b.to(dev))
    Lg = float(ops.loss_value(sums, C * H * W * 3, 0.2))
    v = ops.loss_bwd(a.to(dev), b.to(dev), scratch, 0.2, 1.0)
    assert abs(Lg - float(L)) < 1e-5 * max(1.0, abs(float(L)))
    assert rel_err(v.cpu(), ad.grad) < 1e-4


def test_projection_backward_clears_the_gradient_records_it_read(dev):
    """MI3DGS_FLAG_CLEAR_VSPLATS: same gradients, and afterwards the whole v_splats buffer is zero -- rasterize_bwd writes rows of
    visible Gaussians only, the projection backward reads exactly those."""
    ops = _ops()
    sc = small_scene(n=700, seed=31, big=True, width=96, height=64, n_views=1)
    sc.params["means"][:100] *= 5.0           # some outside the frustum
    P = {k: v.to(dev) for k, v in sc.params.items()}
    vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
    fl = ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC
    radii, splats = ops.project_fwd(P["means"], P["quats"], P["scales"], P["opacities"], vm, K, sc.width, sc.height, sh0=P["sh0"], shN=P["shN"],
                                    sh_degree=3, flags=fl)
    b = ops.bin_tiles(radii, splats, sc.width, sc.height, 16)
    r, a, l = ops.rasterize_fwd(splats, b, sc.width, sc.height, 16)
    g = torch.Generator().manual_seed(1)
    vr, va = torch.randn(r.shape, generator=g).to(dev), torch.randn(a.shape, generator=g).to(dev)
    outs = []
    for clear in (0, ops.FLAG_CLEAR_VSPLATS):
        vs = ops.rasterize_bwd(splats, b, sc.width, sc.height, a, l, vr, va, 16)
        vis = (radii > 0).all(-1)[0]
        assert float(vs[0][~vis].abs().sum()) == 0.0 and float(vs[0][vis].abs().sum()) > 0
        out = ops.project_bwd(P["means"], P["quats"], P["scales"], P["opacities"], vm, K, sc.width, sc.height, radii, splats, vs, sh0=P["sh0"],
                              shN=P["shN"], color_mode=ops.COLOR_SH, sh_degree=3, flags=fl | clear)
        outs.append({k: v.clone() for k, v in out.items() if torch.is_tensor(v)})
        assert (float(vs.abs().sum()) == 0.0) == bool(clear)
    for k in outs[0]:
        assert rel_err(outs[1][k], outs[0][k]) < 1e-5, k          # (two rasterize_bwd runs: float atomics sum in another order)


def test_loss_reads_the_uint8_image_cache_itself(dev):
    """Same bits as converting first (mi3dgs_image_u8_to_f32, value * (1 / 255)) and calling the float32 entry points."""
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    C, H, W = 2, 77, 101
    r = torch.rand(C, H, W, 3, generator=g).to(dev)
    t8 = torch.randint(0, 256, (C, H, W, 3), generator=g, dtype=torch.uint8).to(dev)
    tf = ops.image_u8_to_f32(t8)
    assert torch.equal(tf, t8.float() * torch.tensor(1.0 / 255.0, device=dev))
    s_f, sc_f = ops.loss_fwd(r, tf)
    v_f = ops.loss_bwd(r, tf, sc_f, 0.2, 1.0).clone()
    s_f, dm_f = s_f.clone(), [sc_f[k].clone() for k in ("dm1", "dm2", "dm3")]
    s_8, sc_8 = ops.loss_fwd(r, t8)
    v_8 = ops.loss_bwd(r, t8, sc_8, 0.2, 1.0)
    assert all(torch.equal(a, sc_8[k]) for a, k in zip(dm_f, ("dm1", "dm2", "dm3"))) and torch.equal(v_f, v_8)
    assert torch.allclose(s_f, s_8, rtol=1e-6)          # (block sums meet in float atomics: order)


# -------------------------------------------------------------------------------- Adam
def test_adam_matches_oracle_and_torch(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    shapes = [(1000, 3), (1000, 4), (1000, 3), (1000, 1), (1000, 3), (1000, 45)]
    lrs = [1.6e-4, 1e-3, 5e-3, 5e-2, 2.5e-3, 1.25e-4]
    P = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.double().clone() for p in P]
    M = [torch.zeros_like(p) for p in ref]
    V = [torch.zeros_like(p) for p in ref]
    gp = [p.to(dev) for p in P]
    gm = [torch.zeros_like(p) for p in gp]
    gv = [torch.zeros_like(p) for p in gp]
    tp = [torch.nn.Parameter(p.clone()) for p in P]
    opt = torch.optim.Adam([{"params": [t], "lr": lr} for t, lr in zip(tp, lrs)], eps=1e-15)
    for step in range(1, 6):
        G = [torch.randn(s, generator=g) * (10.0 ** float(torch.randint(-6, 0, (1,), generator=g))) for s in shapes]
        for i in range(6):
            ref[i], M[i], V[i] = O.adam_step(ref[i], G[i].double(), M[i], V[i], step, lrs[i])
            tp[i].grad = G[i].clone()
        opt.step()
        # only the first 900 rows are live: the tail must stay untouched
        ops.adam_step(gp, [x.to(dev) for x in G], gm, gv, lrs, step, numel=[900 * s[1] for s in shapes])
    for i in range(6):
        assert rel_err(gp[i][:900].cpu(), ref[i][:900]) < 1e-6
        assert rel_err(gp[i][:900].cpu(), tp[i].detach()[:900]) < 1e-6
        assert torch.equal(gp[i][900:].cpu(), P[i][900:])


# ----------------------------------------------------------------------------- densify
@pytest.mark.parametrize("step,expect_screen", [(3100, True), (2500, True), (4000, False)])
def test_densify_screen_size_rules_match_the_oracle_masks(dev, step, expect_screen):
    """`ns-train splatfacto` (reference main.py:1270-1306): split_screen_size 0.05 / cull_screen_size 0.15 until step 4000 =
    gsplat's grow_scale2d / prune_scale2d / refine_scale2d_stop_iter.  Decisions against oracle.strategy_masks, including
    Gaussians that are duplicated AND split (three outputs), the prune half only once step > reset_every, and nothing of it
    from the stop iteration on."""
    from mi3dgs import trainer
    n = 1000
    sc = small_scene(n=n, seed=12)
    sc.params["scales"][:300] = math.log(0.005)      # below grow_scale3d: duplicate candidates
    sc.params["opacities"][::19] = -6.0              # below prune_opa
    g = sc.to(dev)
    img = torch.rand(2, sc.height, sc.width, 3, device=dev)
    cfg = trainer.TrainConfig(capacity=4000, reset_every=3000, grow_scale2d=0.05, prune_scale2d=0.15, refine_scale2d_stop_iter=4000,
                              prune_scale3d=0.5)
    tr = trainer.Trainer(g.params, g.viewmats, g.Ks, img, sc.width, sc.height, cfg)
    assert "radii" in tr.stats
    gen = torch.Generator().manual_seed(6)
    tr.stats["grad2d"][:n] = (torch.rand(n, generator=gen) * 6e-4).to(dev)
    tr.stats["count"][:n] = torch.randint(0, 3, (n,), generator=gen).float().to(dev)
    tr.stats["radii"][:n] = (torch.rand(n, generator=gen) * 0.2).to(dev)        # a quarter above 0.15, three quarters above 0.05
    for gname in trainer.GROUPS:
        tr.model.state(gname, "m").fill_(1.0)
    P = {k: v.clone() for k, v in sc.params.items()}
    state = {k: tr.stats[k][:n].cpu().double() for k in ("grad2d", "count", "radii")}
    # the statistic is stored in float32; compare the oracle's thresholds on the same values
    tr.step_count = step
    sc_exp, op_sig = P["scales"].double().exp(), torch.sigmoid(P["opacities"].double())
    dup, split, _ = O.strategy_masks(state, sc_exp, op_sig, step, prune_scale3d=0.5, grow_scale2d=float(np.float32(0.05)),
                                     prune_scale2d=float(np.float32(0.15)), refine_scale2d_stop_iter=4000)
    dup0, split0, _ = O.strategy_masks(state, sc_exp, op_sig, step, prune_scale3d=0.5)
    assert bool((split & ~split0).any()) == expect_screen and torch.equal(dup, dup0)
    if expect_screen:
        assert int((dup & split).sum()) > 10             # the three-output case is exercised
    # prune on the grown set: samples of a split are 1.6 x smaller, every child carries the parent's radius statistic
    s_eff = sc_exp.amax(-1) / torch.where(split & ~dup, 1.6, 1.0)
    prune = op_sig < 0.005
    if step > 3000:
        prune = prune | (s_eff > 0.5)
        if step < 4000:
            prune = prune | (state["radii"] > float(np.float32(0.15)))
    info = tr.refine(do_grow=True)
    flags = tr.flags_buf[:n].cpu()
    assert torch.equal((flags & 1).bool(), dup) and torch.equal((flags & 2).bool(), split) and torch.equal((flags & 4).bool(), prune)
    copies = (~prune).long() * (1 + dup.long() + split.long())
    assert info["n_after"] == int(copies.sum()) == tr.model.n
    assert (info["n_dup"] == int((dup & ~prune).sum()) and info["n_split"] == int((split & ~prune).sum())
            and info["n_prune"] == int(prune.sum()))
    offs = torch.cumsum(copies, 0) - copies
    means, scales = tr.model.p("means").cpu(), tr.model.p("scales").cpu()
    m_state = tr.model.state("means", "m").cpu()
    # duplicated AND split: [untouched copy, sample, sample], all three with zero Adam state
    both = dup & split & ~prune
    if expect_screen:
        b = offs[both]
        assert torch.equal(means[b], P["means"][both]) and torch.equal(scales[b], P["scales"][both])
        for k in (1, 2):
            assert torch.allclose(scales[b + k], P["scales"][both] - math.log(1.6), atol=1e-6)
            assert float((means[b + k] - P["means"][both]).norm(dim=-1).min()) > 0
        assert float((means[b + 1] - means[b + 2]).norm(dim=-1).min()) > 0
        assert float(torch.cat([m_state[b], m_state[b + 1], m_state[b + 2]]).abs().max()) == 0.0
    # split only: two samples; duplicate only: original keeps its state, the copy starts from zero
    so = offs[split & ~dup & ~prune]
    assert torch.allclose(scales[so], P["scales"][split & ~dup & ~prune] - math.log(1.6), atol=1e-6)
    assert float(m_state[so].abs().max()) == 0.0 and float(m_state[so + 1].abs().max()) == 0.0
    do = offs[dup & ~split & ~prune]
    assert torch.equal(means[do], means[do + 1]) and float(m_state[do].min()) == 1.0 and float(m_state[do + 1].abs().max()) == 0.0
    # untouched survivors bit for bit, in order
    plain = ~prune & ~dup & ~split
    for gname, w in zip(trainer.GROUPS, trainer.WIDTHS):
        assert torch.equal(tr.model.banks[tr.model.cur][gname]["p"][: tr.model.n].cpu()[offs[plain]], P[gname].reshape(n, w)[plain])
    assert float(tr.stats["radii"].abs().max()) == 0.0


def test_screen_radius_statistic_is_the_running_max_and_only_written_while_read(dev):
    """gsplat DefaultStrategy._update_state: state['radii'] = max(state['radii'], radii / max(W, H)) over the steps since the
    last refine, visible Gaussians only; a preset without the rule allocates no such array, and from the stop iteration on the
    projection backward is not handed it any more."""
    from mi3dgs import ops, trainer
    sc = small_scene(n=700, seed=13)
    g = sc.to(dev)
    img = torch.rand(2, sc.height, sc.width, 3, device=dev)
    off = trainer.Trainer(g.params, g.viewmats, g.Ks, img, sc.width, sc.height, trainer.TrainConfig(capacity=1000))
    assert "radii" not in off.stats
    tr = trainer.Trainer(g.params, g.viewmats, g.Ks, img, sc.width, sc.height,
                         trainer.TrainConfig(capacity=1000, refine_scale2d_stop_iter=3, refine_start_iter=10_000))
    expect = torch.zeros(700)
    for i in range(3):
        tr.step(i % 2)
        r = tr.radii[0, :700].cpu()
        vis = (r > 0).all(-1)
        expect = torch.where(vis, torch.maximum(expect, r.max(-1).values.float() / max(sc.width, sc.height)), expect)
    assert torch.equal(tr.stats["radii"][:700].cpu(), expect) and float(expect.max()) > 0
    tr.step(0)                              # step 3 = the stop iteration: not touched any more
    assert torch.equal(tr.stats["radii"][:700].cpu(), expect)


def test_densify_decisions_and_surgery(dev):
    from mi3dgs import trainer
    sc = small_scene(n=1000, seed=10)
    sc.params["scales"][:300] = math.log(0.005)      # below grow_scale3d: duplicate candidates
    sc.params["opacities"][::17] = -6.0              # below prune_opa
    g = sc.to(dev)
    img = torch.rand(2, sc.height, sc.width, 3, device=dev)
    cfg = trainer.TrainConfig(capacity=3000, reset_every=3000)
    tr = trainer.Trainer(g.params, g.viewmats, g.Ks, img, sc.width, sc.height, cfg)
    gen = torch.Generator().manual_seed(5)
    tr.stats["grad2d"][:1000] = (torch.rand(1000, generator=gen) * 6e-4).to(dev)
    tr.stats["count"][:1000] = torch.randint(0, 3, (1000,), generator=gen).float().to(dev)
    # give every group a recognisable Adam state
    for gname in trainer.GROUPS:
        tr.model.state(gname, "m").fill_(1.0)
        tr.model.state(gname, "v").fill_(2.0)
    P = {k: v.clone() for k, v in sc.params.items()}
    state = dict(grad2d=tr.stats["grad2d"][:1000].cpu().double(), count=tr.stats["count"][:1000].cpu().double())
    tr.step_count = 3100            # > reset_every: the too-big rule is active
    dup, split, prune0 = O.strategy_masks(state, P["scales"].double().exp(), torch.sigmoid(P["opacities"].double()),
                                          3100)
    # prune is evaluated on the grown set: split children have scale / 1.6
    s_eff = P["scales"].double().exp().amax(-1) / torch.where(split, 1.6, 1.0)
    prune = (torch.sigmoid(P["opacities"].double()) < 0.005) | (s_eff > 0.1)
    info = tr.refine(do_grow=True)
    exp_n = int(((~prune) * (1 + (dup | split).long())).sum())
    assert info["n_after"] == exp_n == tr.model.n
    assert info["n_dup"] == int((dup & ~prune).sum()) and info["n_split"] == int((split & ~prune).sum())
    assert info["n_dup"] > 0 and info["n_split"] > 0 and info["n_prune"] == int(prune.sum())
    # survivors keep order; untouched ones keep params and Adam state bit for bit
    keep_plain = (~prune) & ~dup & ~split
    offs = torch.cumsum((~prune).long() * (1 + (dup | split).long()), 0) - (~prune).long() * (1 + (dup | split).long())
    idx = offs[keep_plain]
    for gname, w in zip(trainer.GROUPS, trainer.WIDTHS):
        got = tr.model.banks[tr.model.cur][gname]["p"][: tr.model.n].cpu()
        assert torch.equal(got[idx], P[gname].reshape(1000, w)[keep_plain])
        assert torch.equal(tr.model.state(gname, "m").cpu()[idx], torch.ones(len(idx), w))
    # duplicates: original keeps state, copy has zero state and identical params
    d_idx = offs[dup & ~prune]
    means = tr.model.p("means").cpu()
    assert torch.equal(means[d_idx], means[d_idx + 1])
    assert float(tr.model.state("means", "m").cpu()[d_idx].min()) == 1.0
    assert float(tr.model.state("means", "m").cpu()[d_idx + 1].abs().max()) == 0.0
    # splits: two children, scale - log 1.6, zero state, displaced means
    s_idx = offs[split & ~prune]
    sc_new = tr.model.p("scales").cpu()
    assert torch.allclose(sc_new[s_idx], P["scales"][split & ~prune] - math.log(1.6), atol=1e-6)
    assert torch.allclose(sc_new[s_idx + 1], sc_new[s_idx])
    assert float(tr.model.state("scales", "v").cpu()[s_idx].abs().max()) == 0.0
    disp = (means[s_idx] - P["means"][split & ~prune]).norm(dim=-1)
    assert float(disp.min()) > 0 and float((means[s_idx] - means[s_idx + 1]).norm(dim=-1).min()) > 0
    # displacement is a sample of N(0, R S^2 R^T): |disp| / max_scale should be O(1)
    ratio = disp / P["scales"][split & ~prune].exp().amax(-1)
    assert 0.3 < float(ratio.median()) < 3.0
    # statistics are reset
    assert float(tr.stats["grad2d"].abs().max()) == 0.0
    # opacity reset
    tr.reset_opacity()
    thr = math.log(0.01 / 0.99)
    assert float(tr.model.p("opacities").max()) <= thr + 1e-6
    assert float(tr.model.state("opacities", "m").abs().max()) == 0.0


def test_scale_reg_matches_oracle(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    s = (torch.randn(500, 3, generator=g) * 1.5).double().requires_grad_(True)
    L = O.scale_regularisation(s.exp())
    L.backward()
    v = torch.zeros(500, 3, device=dev)
    ls = torch.zeros(1, device=dev)
    ops.scale_reg(s.detach().float().to(dev), 0.1, 10.0, v, ls)
    assert abs(float(ls) - float(L)) < 1e-4 * float(L)
    assert rel_err(v.cpu(), s.grad) < 1e-4


# ---------------------------------------------------------------- end to end: training
def test_training_recovers_target(dev):
    """Render targets from one Gaussian set, perturb it, train: the loss must fall steadily
    (end-to-end check in lieu of a reference golden image)."""
    from mi3dgs import trainer
    sc = small_scene(n=1500, seed=11, big=True, width=96, height=64, n_views=4, fx=90.0)
    g = sc.to(dev)
    tr0 = trainer.Trainer(g.params, g.viewmats, g.Ks, torch.zeros(4, 64, 96, 3, device=dev), 96, 64,
                          trainer.TrainConfig(densify=False))
    imgs = torch.cat([tr0.render(g.viewmats[i], g.Ks[i])[0].clone() for i in range(4)])
    gen = torch.Generator().manual_seed(7)
    P = {k: v.clone() for k, v in g.params.items()}
    P["means"] = P["means"] + 0.02 * torch.randn(1500, 3, generator=gen).to(dev)
    P["sh0"] = P["sh0"] + 0.3 * torch.randn(1500, 1, 3, generator=gen).to(dev)
    cfg = trainer.TrainConfig(max_steps=300, densify=False, sh_degree_interval=1)
    tr = trainer.Trainer(P, g.viewmats, g.Ks, imgs, 96, 64, cfg)
    first = sum(tr.step(i % 4, want_loss=True) for i in range(4)) / 4
    for i in range(4, 296):
        tr.step(i % 4)
    last = sum(tr.step(i % 4, want_loss=True) for i in range(4)) / 4
    assert last < 0.5 * first, (first, last)


def test_culled_groups_adam_on_a_second_stream_is_the_same_update(dev):
    """TrainConfig.overlap_culled_adam: the fused backward + Adam as two launches (fully culled 64-Gaussian groups on a side stream
    after the projection, the others after the rasteriser) updates every Gaussian exactly as the one launch does."""
    from mi3dgs import trainer
    sc = small_scene(n=6000, seed=13, big=False, width=96, height=64, n_views=4, fx=90.0)
    sc.params["means"][:2500] += torch.tensor([30.0, 0.0, 0.0])          # a good part of the scene outside every view
    g = sc.to(dev)
    tr0 = trainer.Trainer(g.params, g.viewmats, g.Ks, torch.zeros(4, 64, 96, 3, device=dev), 96, 64, trainer.TrainConfig(densify=False))
    imgs = torch.cat([tr0.render(g.viewmats[i], g.Ks[i])[0].clone() for i in range(4)])
    gen = torch.Generator().manual_seed(9)
    P = {k: v.clone() for k, v in g.params.items()}
    P["sh0"] = P["sh0"] + 0.2 * torch.randn(6000, 1, 3, generator=gen).to(dev)
    out = []
    for mode in (None, "after_project", "after_binning"):
        cfg = trainer.TrainConfig(max_steps=300, densify=False, sh_degree_interval=1, spatial_sort_init=True, overlap_culled_adam=mode,
                                  overlap_min_gaussians=0, use_scale_regularization=True, scale_reg_every=3)
        tr = trainer.Trainer(P, g.viewmats, g.Ks, imgs, 96, 64, cfg)
        for i in range(12):
            tr.step(i % 4)
        torch.cuda.synchronize()
        assert bool(tr._overlap_on) == (mode is not None)          # (the two-launch path really ran)
        vis = (tr.radii[0, :6000] > 0).all(-1)
        out.append(({k: tr.model.p(k).clone() for k in trainer.GROUPS}, {k: tr.model.state(k, "m").clone() for k in trainer.GROUPS}, vis))
    assert 64 * 20 < int((~out[0][2]).sum()) < 6000 - 64 * 20
    for k in trainer.GROUPS:
        for o in out[1:]:
            assert rel_err(o[0][k], out[0][0][k]) < 2e-3 and rel_err(o[1][k], out[0][1][k]) < 2e-3, k


@pytest.mark.parametrize("sreg", [0.0, 0.1])
def test_culled_groups_kernel_gives_the_fused_kernels_bits(dev, sreg):
    """(sreg: in a step with splatfacto's scale regulariser a culled Gaussian has that one gradient, formed by both kernels.)
    mi3dgs_adam_culled_groups against mi3dgs_project_bwd_adam restricted to the same groups (MI3DGS_FLAG_ONLY_CULLED_GROUPS): random
    parameters and moments, groups of 64 marked culled / visible through the radii -- bit for bit on the culled groups, nothing
    touched elsewhere; including a last group of fewer than 64 Gaussians."""
    ops = _ops()
    from mi3dgs import trainer
    g = torch.Generator().manual_seed(14)
    N = 64 * 37 + 23
    W = dict(zip(trainer.GROUPS, trainer.WIDTHS))
    lrs = [1.6e-4, 1e-3, 5e-3, 5e-2, 2.5e-3, 1.25e-4]
    radii = torch.zeros(1, N, 2, dtype=torch.int32)
    vis_groups = torch.rand(38, generator=g) < 0.5
    vis_groups[37] = False                                 # the partial last group is culled
    for gi in range(38):
        if vis_groups[gi]:
            radii[0, 64 * gi + int(torch.randint(0, 64, (1,), generator=g)) % max(1, min(64, N - 64 * gi))] = 3      # ONE visible member
    radii = radii.to(dev)
    culled = ~vis_groups.repeat_interleave(64)[:N].to(dev)

    def fresh():
        gg = torch.Generator().manual_seed(15)
        P = [torch.randn(N, W[k], generator=gg).to(dev) for k in trainer.GROUPS]
        M = [(0.01 * torch.randn(N, W[k], generator=gg)).to(dev) for k in trainer.GROUPS]
        V = [(1e-4 * torch.rand(N, W[k], generator=gg)).to(dev) for k in trainer.GROUPS]
        return P, M, V

    P0, M0, V0 = fresh()
    P1, M1, V1 = fresh()
    P2, M2, V2 = fresh()
    ops.adam_culled_groups(P1, M1, V1, lrs, 7, radii, n=N, scale_reg_weight=sreg, scale_reg_max_ratio=10.0)
    vm = torch.eye(4, device=dev)[None].contiguous()
    K = torch.tensor([[[50.0, 0, 32], [0, 50.0, 24], [0, 0, 1]]], device=dev)
    splats = torch.zeros(1, N, ops.SPLAT_STRIDE, device=dev)
    v_splats = torch.zeros(1, N, ops.GRAD_STRIDE, device=dev)
    ops.project_bwd_adam(P2, M2, V2, lrs, 7, vm, K, 64, 48, radii, splats, v_splats, n=N, sh_degree=3,
                         flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC | ops.FLAG_ONLY_CULLED_GROUPS, scale_reg_weight=sreg,
                         scale_reg_max_ratio=10.0)
    if sreg:        # the regulariser is active on a good part of them (log-scales ~ N(0, 1): ratio = exp(max - min))
        sc0 = P0[2]
        assert 0.2 < float(((sc0.amax(1) - sc0.amin(1)).exp() > 10.0).float().mean()) < 0.9
    assert int(culled.sum()) > 64 * 8 and int((~culled).sum()) > 64 * 8
    for a, b, z in zip(P1 + M1 + V1, P2 + M2 + V2, P0 + M0 + V0):
        assert torch.equal(a[culled], b[culled])                 # the two kernels: the same bits
        assert torch.equal(a[~culled], z[~culled]) and torch.equal(b[~culled], z[~culled])      # the other groups: untouched
        assert not torch.equal(a[culled], z[culled])


def test_morton_ordered_start_trains_the_same_gaussians(dev):
    """TrainConfig.spatial_sort_init permutes the initial Gaussians along a Morton curve: the same training, Gaussian for Gaussian,
    up to the order in which float atomics meet (including a refine pass, which keeps children next to their parents)."""
    from mi3dgs import trainer
    sc = small_scene(n=3000, seed=12, big=False, width=96, height=64, n_views=4, fx=90.0)
    g = sc.to(dev)
    tr0 = trainer.Trainer(g.params, g.viewmats, g.Ks, torch.zeros(4, 64, 96, 3, device=dev), 96, 64, trainer.TrainConfig(densify=False))
    imgs = torch.cat([tr0.render(g.viewmats[i], g.Ks[i])[0].clone() for i in range(4)])
    gen = torch.Generator().manual_seed(8)
    P = {k: v.clone() for k, v in g.params.items()}
    P["means"] = P["means"] + 0.02 * torch.randn(3000, 3, generator=gen).to(dev)
    perm = trainer.morton_order(P["means"]).to(dev)
    assert sorted(perm.tolist()) == list(range(3000)) and not torch.equal(perm, torch.arange(3000, device=dev))
    out = []
    for sort in (False, True):
        cfg = trainer.TrainConfig(max_steps=300, densify=False, sh_degree_interval=1, spatial_sort_init=sort)
        tr = trainer.Trainer(P, g.viewmats, g.Ks, imgs, 96, 64, cfg)
        losses = [tr.step(i % 4, want_loss=True) for i in range(24)]
        out.append((losses, {k: tr.model.p(k).clone() for k in trainer.GROUPS}))
    (l0, p0), (l1, p1) = out
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-5 * max(l0)
    for k in trainer.GROUPS:          # (Adam turns the last bits of a small gradient into full-size steps: the bound of the other
        assert rel_err(p1[k], p0[k][perm]) < 2e-3, k      #  trainer-equivalence tests; 3.5e-4 measured on the opacities after 24 steps)


@pytest.mark.parametrize("two_cams", [False, True])
@pytest.mark.parametrize("vgrad", [0.0, 1e-3])
def test_projection_backward_stays_finite_on_a_needle_gaussian(dev, two_cams, vgrad):
    """Found in the MCMC synthetic run (tools/nan_probe.py): a needle-thin Gaussian 0.09 in front of the camera, projecting
    thousands of pixels wide, has a 2D determinant that is all rounding; the training kernel recomputed it, got 0 where the
    forward had got a small positive number, and 0 * inf made every geometric gradient NaN even with an all-zero gradient
    record -- for good, since Adam keeps a NaN.  The backward now takes the conic the forward stored."""
    ops = _ops()
    means = torch.tensor([[-0.5355250835418701, -0.0883798897266388, -1.2471290826797485]], device=dev)
    quats = torch.tensor([[1.1715995073318481, 1.1246978044509888, 0.7527013421058655, 0.2929958701133728]], device=dev)
    scales = torch.tensor([[-17.504308700561523, -0.5044186115264893, -3.654362201690674]], device=dev)
    opac = torch.tensor([-3.7583377361297607], device=dev)
    sh0, shN = torch.zeros(1, 1, 3, device=dev), torch.zeros(1, 15, 3, device=dev)
    vm = torch.tensor([[[0.07606703788042068, 0.9971022605895996, 0.0009645117679610848, 0.0006943568005226552],
                        [-0.3021939992904663, 0.022131966426968575, 0.9529894590377808, -0.0068558864295482635],
                        [0.9502065777778625, -0.07278256118297577, 0.3030018210411072, 0.9749422073364258],
                        [0.0, 0.0, 0.0, 1.0]]], device=dev)
    K = torch.tensor([[[725.0, 0, 480.0], [0, 725.0, 270.0], [0, 0, 1]]], device=dev)
    if two_cams:                                   # the generic (any number of cameras) kernel
        vm, K = vm.repeat(2, 1, 1), K.repeat(2, 1, 1)
    radii, splats = ops.project_fwd(means, quats, scales, opac, vm, K, 960, 540, sh0=sh0, shN=shN, sh_degree=3, flags=3)
    assert bool((radii > 1000).all())              # visible, and enormous
    v = torch.full((vm.shape[0], 1, 16), vgrad, device=dev)
    out = ops.project_bwd(means, quats, scales, opac, vm, K, 960, 540, radii, splats, v, sh0=sh0, shN=shN,
                          color_mode=ops.COLOR_SH, sh_degree=3, flags=3)
    for k, t in out.items():
        assert bool(torch.isfinite(t).all()), k
    if vgrad == 0.0:
        assert all(float(out[k].abs().max()) == 0.0 for k in ("v_means", "v_quats", "v_scales"))
    else:
        assert float(out["v_means"].abs().max()) > 0


def test_auto_isect_capacity_follows_the_exact_path_and_grows_on_overflow(dev):
    """TrainConfig.auto_isect_capacity (what the ns-train / simple_trainer shims run with): tile-list buffers sized from
    measured counts so that no step reads the count back.  Same training as the exact (count read back every step) path up
    to float-atomic order; a capacity made too small on purpose is reported, not fatal, and grows."""
    import dataclasses
    from mi3dgs import trainer
    sc = small_scene(n=1500, seed=12, big=True, width=96, height=64, n_views=4, fx=90.0)
    g = sc.to(dev)
    tr0 = trainer.Trainer(g.params, g.viewmats, g.Ks, torch.zeros(4, 64, 96, 3, device=dev), 96, 64,
                          trainer.TrainConfig(densify=False))
    imgs = torch.cat([tr0.render(g.viewmats[i], g.Ks[i])[0].clone() for i in range(4)]) * 0.8 + 0.1
    cfg = trainer.TrainConfig(max_steps=200, sh_degree_interval=1, refine_start_iter=5, refine_every=10, capacity=6000)
    P0 = {k: v.clone() for k, v in g.params.items()}
    tr_x = trainer.Trainer({k: v.clone() for k, v in P0.items()}, g.viewmats, g.Ks, imgs, 96, 64, cfg)
    tr_a = trainer.Trainer({k: v.clone() for k, v in P0.items()}, g.viewmats, g.Ks, imgs, 96, 64,
                           dataclasses.replace(cfg, auto_isect_capacity=True))
    for i in range(40):
        tr_x.step(i % 4)
        tr_a.step(i % 4)
    tr_a.check_async_errors()
    assert tr_a._auto_cap is not None and tr_a.isect_overflows == 0
    assert tr_a.last_binning["max_isect"] == tr_a.last_binning["flatten_ids"].numel() > 0      # the capacity path ran
    assert tr_a.model.n == tr_x.model.n > 1500                                                  # same refine decisions
    for grp in ("means", "scales", "opacities", "sh0"):
        a, x = tr_a.model.p(grp)[: tr_a.model.n], tr_x.model.p(grp)[: tr_x.model.n]
        assert rel_err(a, x) < 1e-3, grp
    # far too small: the emit pass clamps, the sticky word says so, the trainer grows instead of raising
    tr_a._auto_cap = 256
    tr_a.step(0)
    tr_a.check_async_errors()
    assert tr_a.isect_overflows == 1 and tr_a._auto_cap >= 512
    tr_a.calibrate_isect_capacity()
    tr_a.step(1)
    tr_a.check_async_errors()
    assert tr_a.isect_overflows == 1


# ------------------------------------------------------------- exact ("tight") tile culling
@pytest.mark.parametrize("seed,big", [(41, True), (42, False), (43, True)])
def test_tight_binning_is_conservative_and_renders_bit_identical(dev, seed, big):
    ops = _ops()
    sc = small_scene(n=900, seed=seed, big=big, n_views=2, width=112, height=72)
    # elongated splats at odd angles are the interesting case for an ellipse-vs-tile test
    sc.params["scales"][:300, 0] += 1.5
    sc.params["scales"][:300, 1] -= 1.0
    g = sc.to(dev)
    radii, splats = ops.project_fwd(g.params["means"], g.params["quats"], g.params["scales"], g.params["opacities"],
                                    g.viewmats, g.Ks, sc.width, sc.height, sh0=g.params["sh0"], shN=g.params["shN"],
                                    sh_degree=3, flags=3)
    box = ops.bin_tiles(radii, splats, sc.width, sc.height, 16, want_tiles_per_gauss=True)
    tight = ops.bin_tiles(radii, splats, sc.width, sc.height, 16, want_tiles_per_gauss=True, tight=True)
    N = 900

    def codes(b):
        return torch.sort(b["tile_keys"].long().cpu() * N + (b["flatten_ids"].long().cpu() % N)
                          + 0 * b["flatten_ids"].long().cpu()).values

    def pair_codes(b):
        fid = b["flatten_ids"].long().cpu()
        return torch.sort(b["tile_keys"].long().cpu() * N + fid % N).values

    sp = splats.cpu()
    need = O.contributing_pairs(sp[..., 0:2], sp[..., 2:5], sp[..., 5], radii.cpu(), sc.width, sc.height)
    cb, ct = pair_codes(box), pair_codes(tight)
    n_box, n_tight = cb.numel(), ct.numel()
    assert n_tight < n_box and need.numel() <= n_tight
    assert torch.isin(need, ct).all(), "tight binning dropped a contributing (tile, splat) pair"
    assert torch.isin(ct, cb).all(), "tight binning invented a pair outside the bounding box"
    assert int(tight["n_isect"].item()) == n_tight == int(tight["tiles_per_gauss"].sum())
    # per-tile lists stay depth-ordered and the offsets are consistent
    tk = tight["tile_keys"].long().cpu()
    assert (tk[1:] >= tk[:-1]).all()
    offs = tight["isect_offsets"].flatten().long().cpu()
    assert torch.equal(offs, torch.searchsorted(tk, torch.arange(offs.numel())))
    dep = sp[..., 9].flatten()[tight["flatten_ids"].long().cpu()]
    same_tile = tk[1:] == tk[:-1]
    assert (dep[1:][same_tile] >= dep[:-1][same_tile]).all()
    # renders: bit for bit
    bg = torch.rand(2, 3, device=dev)
    r0, a0, l0 = [t.clone() for t in ops.rasterize_fwd(splats, box, sc.width, sc.height, 16, bg)]
    r1, a1, l1 = ops.rasterize_fwd(splats, tight, sc.width, sc.height, 16, bg)
    assert torch.equal(r0, r1) and torch.equal(a0, a1)
    # gradients: same sums (float atomics order differs), so compare closely
    gen = torch.Generator().manual_seed(1)
    vr = torch.randn(r0.shape, generator=gen).to(dev)
    va = torch.randn(a0.shape, generator=gen).to(dev)
    g0 = ops.rasterize_bwd(splats, box, sc.width, sc.height, a0, l0, vr, va, 16, bg)
    g1 = ops.rasterize_bwd(splats, tight, sc.width, sc.height, a1, l1, vr, va, 16, bg)
    assert rel_err(g1, g0) < 1e-5
    print(f"intersections: box {n_box}, tight {n_tight} ({100.0 * n_tight / n_box:.0f} %), exact {need.numel()}")


@pytest.mark.parametrize("fused", [False, True])
def test_big_splats_take_the_wave_per_splat_emit_path(dev, fused):
    """Splats tens of tile rows tall overflow the block's row table (2 048 rows per 256 splats) and are emitted by
    the wave-per-splat path of tile_emit: same pair set rules as everywhere else, renders identical to box binning."""
    ops = _ops()
    W, H = 352, 288                                           # 22 x 18 tiles
    sc = small_scene(n=700, seed=61, big=True, n_views=1, width=W, height=H, fx=200.0)
    sc.params["scales"] += 1.6                                # radii of a hundred pixels and more
    sc.params["scales"][:400, 0] += 1.0
    g = sc.to(dev)
    radii, splats = ops.project_fwd(g.params["means"], g.params["quats"], g.params["scales"], g.params["opacities"],
                                    g.viewmats, g.Ks, W, H, sh0=g.params["sh0"], shN=g.params["shN"], sh_degree=3, flags=3)
    rows = ((radii[0, :, 1].float() * 2 + 15) / 16).clamp(max=18)
    assert float(rows[radii[0, :, 1] > 0].mean()) > 10.0       # way past 8 rows per splat: the table overflows
    box = ops.bin_tiles(radii, splats, W, H, 16, want_tiles_per_gauss=True)
    I_box = int(box["n_isect"].item())
    tight = ops.bin_tiles(radii, splats, W, H, 16, want_tiles_per_gauss=True, tight=True,
                          max_isect=I_box if fused else None, fused=fused)
    I = int(tight["n_isect"].item())
    assert 0 < I < I_box and I == int(tight["tiles_per_gauss"].sum())
    N = 700
    sp = splats.cpu()
    need = O.contributing_pairs(sp[..., 0:2], sp[..., 2:5], sp[..., 5], radii.cpu(), W, H)
    ct = torch.sort(tight["tile_keys"][:I].long().cpu() * N + tight["flatten_ids"][:I].long().cpu() % N).values
    cb = torch.sort(box["tile_keys"].long().cpu() * N + box["flatten_ids"].long().cpu() % N).values
    assert torch.unique(ct).numel() == ct.numel(), "a (tile, splat) pair was emitted twice"
    assert torch.isin(need, ct).all() and torch.isin(ct, cb).all()
    tk = tight["tile_keys"][:I].long().cpu()
    assert (tk[1:] >= tk[:-1]).all()
    dep = sp[..., 9].flatten()[tight["flatten_ids"][:I].long().cpu()]
    same = tk[1:] == tk[:-1]
    assert (dep[1:][same] >= dep[:-1][same]).all()
    r0, a0, _ = [t.clone() for t in ops.rasterize_fwd(splats, box, W, H, 16)]
    r1, a1, _ = ops.rasterize_fwd(splats, tight, W, H, 16)
    assert torch.equal(r0, r1) and torch.equal(a0, a1)


# --------------------------------------------------------- Adam fused into the backward
@pytest.mark.parametrize("n,scale_reg", [(700, False), (333, True), (64, True)])
def test_fused_adam_backward_equals_separate_kernels(dev, n, scale_reg):
    """Same step through project_bwd + (scale_reg) + adam_step and through project_bwd_adam."""
    from mi3dgs import trainer
    sc = small_scene(n=n, seed=51, big=True, width=80, height=56, n_views=3)
    sc.params["scales"][: n // 3, 0] += 2.5           # ratios above max_gauss_ratio: the regulariser is active
    g = sc.to(dev)
    imgs = torch.rand(3, 56, 80, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    out = {}
    for fuse in (False, True):
        cfg = trainer.TrainConfig(max_steps=50, densify=True, refine_start_iter=10 ** 9, sh_degree_interval=1,
                                  use_scale_regularization=scale_reg, scale_reg_every=1, fuse_adam=fuse, capacity=n + 40)
        tr = trainer.Trainer({k: v.clone() for k, v in g.params.items()}, g.viewmats, g.Ks, imgs, 80, 56, cfg)
        for s in range(5):
            tr.step(s % 3)
        out[fuse] = ({k: tr.model.p(k).clone() for k in trainer.GROUPS},
                     {k: tr.model.state(k, "m").clone() for k in trainer.GROUPS},
                     {k: tr.model.state(k, "v").clone() for k in trainer.GROUPS},
                     {k: v[:n].clone() for k, v in tr.stats.items()},
                     {k: tr.model.banks[0][k]["p"][n:].clone() for k in trainer.GROUPS})
    for part in range(4):
        for k in out[True][part]:
            a, b = out[True][part][k], out[False][part][k]
            # two separate runs: rasterize_bwd's float atomics sum in a different order each time,
            # and 5 Adam steps (eps 1e-15) amplify that; a fusion bug would show at the 1e-1 level
            assert rel_err(a, b) < 1e-3, (part, k, rel_err(a, b))
    # nothing beyond the live Gaussians is touched (capacity tail)
    for k in trainer.GROUPS:
        assert float(out[True][4][k].abs().max()) == 0.0
    # and the step really moved the parameters
    assert rel_err(out[True][0]["means"], g.params["means"]) > 1e-6


def test_fused_backward_reads_nothing_behind_the_last_gaussian(dev):
    """The fused backward + Adam launches whole 256-thread blocks: with 332 Gaussians the second block's last two waves own no
    Gaussian at all.  Their (clamped) loads of the shN parameter and moment arrays once went to index -1 of an offset past the
    arrays' end -- harmless inside an allocator segment, a device fault where a bank ends with its segment (a test order that
    did that took the process down).  Here every array ends exactly at the end of an allocation of its own."""
    from mi3dgs import ops, trainer
    n = 332                                           # 332 * 45 floats is a multiple of 16 bytes: the views below stay aligned
    sc = small_scene(n=n, seed=52, big=True, width=80, height=56, n_views=1)
    g = sc.to(dev)

    def at_end(t):                                    # the tensor's values in the LAST bytes of a 12 MiB allocation (a segment of its own)
        buf = torch.zeros((12 << 20) // 4, dtype=torch.float32, device=dev)
        v = buf[buf.numel() - t.numel():].view(t.shape)
        v.copy_(t)
        return v, buf

    keep = []
    params, m1, m2 = [], [], []
    for k in trainer.GROUPS:
        for lst, src in ((params, g.params[k].float()), (m1, torch.zeros_like(g.params[k]).float()), (m2, torch.zeros_like(g.params[k]).float())):
            v, buf = at_end(src.contiguous())
            lst.append(v); keep.append(buf)
    vm, K = g.viewmats[:1].contiguous(), g.Ks[:1].contiguous()
    radii, splats = ops.project_fwd(params[0], params[1], params[2], params[3], vm, K, 80, 56, sh0=params[4], shN=params[5], sh_degree=3,
                                    flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC)
    v_splats = torch.randn(1, n, ops.GRAD_STRIDE, device=dev) * 1e-3
    before = params[5].clone()
    for sh_degree in (0, 3):
        ops.project_bwd_adam(params, m1, m2, (1e-4, 1e-3, 5e-3, 5e-2, 2.5e-3, 1.25e-4), 1, vm, K, 80, 56, radii, splats, v_splats,
                             n=n, sh_degree=sh_degree, flags=ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(params[5]).all()) and not torch.equal(params[5], before)
    assert ops._lib.async_errors() == 0


# --------------------------------------------------------------------------- MCMC strategy
def test_mcmc_relocation_matches_oracle(dev):
    from mi3dgs import strategy_mcmc
    g = torch.Generator().manual_seed(7)
    n = 300
    op = torch.rand(n, generator=g) * 0.98 + 0.01
    sc = torch.rand(n, 3, generator=g) * 0.1 + 0.001
    ratios = torch.randint(1, 12, (n,), generator=g)
    ratios[:5] = torch.tensor([1, 2, 51, 30, 1])
    no_ref, ns_ref = O.mcmc_relocation(op.double(), sc.double(), ratios)
    no, ns = strategy_mcmc.compute_relocation(op.to(dev), sc.to(dev), ratios.to(dev), strategy_mcmc.binom_table(dev))
    assert rel_err(no.cpu(), no_ref) < 1e-5 and rel_err(ns.cpu(), ns_ref) < 2e-4
    one = ratios == 1
    assert torch.allclose(no.cpu()[one], op[one], atol=1e-6) and torch.allclose(ns.cpu()[one], sc[one], rtol=1e-5)


def test_mcmc_noise_statistics(dev):
    from mi3dgs import ops
    n = 20000
    means = torch.zeros(n, 3, device=dev)
    quats = torch.tensor([[0.9, 0.1, -0.3, 0.2]], device=dev).repeat(n, 1)
    ls = torch.log(torch.tensor([[0.5, 0.1, 0.02]], device=dev)).repeat(n, 1).contiguous()
    opa = torch.full((n,), -9.0, device=dev)          # sigmoid ~ 1e-4: gate ~ 1
    opa[n // 2:] = 3.0                                  # opaque: gate ~ 0
    ops._lib.call("mi3dgs_mcmc_inject_noise", n, ops._p(means), ops._p(quats), ops._p(ls), ops._p(opa), 2.0, 1234,
                  ops._stream(dev))
    moved, still = means[: n // 2].double().cpu(), means[n // 2:]
    assert float(still.abs().max()) < 1e-6
    R = O.quat_to_rotmat(quats[:1].double().cpu())[0]
    Sig = R @ torch.diag(torch.tensor([0.5, 0.1, 0.02], dtype=torch.float64) ** 2) @ R.T
    gate = 1 / (1 + math.exp(-100 * ((1 - 1 / (1 + math.exp(9.0))) - 0.995)))
    cov_ref = (2.0 * gate) ** 2 * Sig @ Sig.T
    cov = (moved.T @ moved) / moved.shape[0]
    assert float((cov - cov_ref).norm() / cov_ref.norm()) < 0.06 and float(moved.mean(0).norm()) < 0.02 * math.sqrt(float(cov_ref.trace()))


def test_mcmc_trainer_relocates_and_grows(dev):
    from mi3dgs import strategy_mcmc, trainer
    sc = small_scene(n=1000, seed=61, big=True, width=96, height=64, n_views=4, fx=90.0)
    sc.params["opacities"][:200] = -8.0               # dead: sigmoid < 0.005
    g = sc.to(dev)
    imgs = torch.rand(4, 64, 96, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    cfg = trainer.TrainConfig(max_steps=100, sh_degree_interval=1)
    tr = strategy_mcmc.MCMCTrainer(g.params, g.viewmats, g.Ks, imgs, 96, 64, cfg,
                                   strategy_mcmc.MCMCConfig(cap_max=1200, refine_start_iter=-1, refine_every=5,
                                                            refine_stop_iter=100))
    before_means = tr.model.p("means").clone()
    n_rel = tr.relocate()
    assert n_rel == 200
    op = torch.sigmoid(tr.model.p("opacities"))
    assert float(op.min()) >= 0.005 - 1e-6                          # nothing dead any more
    moved = (tr.model.p("means")[:200] - before_means[:200]).norm(dim=1)
    assert float(moved.min()) > 0                                   # teleported onto live Gaussians
    same = (tr.model.p("means")[:200, None, :] == before_means[None, 200:, :]).all(-1).any(1)
    assert bool(same.all())                                          # bit-exact copies of live positions
    # growth (gsplat sample_add): the appended copies start from zero moments, the SOURCES keep theirs (relocate
    # zeroes them; doing the same here kicked 5 % of the most opaque Gaussians by ~3 x lr every 100 steps and cost
    # 3.5 dB on the long synthetic run, profiles/r02_train_synthetic.txt)
    for k in trainer.GROUPS:
        tr.model.state(k, "m").fill_(0.25)
        tr.model.state(k, "v").fill_(0.5)
    n_new = tr.add_new()
    assert n_new == 50 and tr.model.n == 1050                       # +5 %
    assert float(tr.model.state("means", "m")[1000:].abs().max()) == 0 and float(tr.model.state("shN", "v")[1000:].abs().max()) == 0
    assert float(tr.model.state("means", "m")[:1000].min()) == 0.25 and float(tr.model.state("opacities", "v")[:1000].min()) == 0.5
    src = (tr.model.p("means")[1000:, None, :] == tr.model.p("means")[None, :1000, :]).all(-1)
    assert bool(src.any(1).all())                                   # every new Gaussian sits on one of the old ones
    tr.model.state("means", "m")[:200] = 0.25
    n_rel2 = tr.relocate()                                          # nothing is dead now
    assert n_rel2 == 0
    for s in range(12):
        tr.step(s % 4)
    assert tr.model.n == 1200                                       # capped at cap_max
    assert all(torch.isfinite(tr.model.p(k)).all() for k in trainer.GROUPS)


def test_mcmc_step_through_the_fused_kernel_equals_the_three_launches(dev):
    """MCMCTrainer: backward + mi3dgs_mcmc_regularise + Adam (fuse_adam=False) against the one fused launch with the regularisers
    folded in (mi3dgs_project_bwd_adam_mcmc), including steps that also apply splatfacto's scale regulariser (those take the
    unfused launches in both runs).  No relocation / noise in these steps (refine_start_iter beyond them, noise_lr 0)."""
    from mi3dgs import trainer
    from mi3dgs.strategy_mcmc import MCMCConfig, MCMCTrainer
    sc = small_scene(n=1500, seed=16, big=True, width=96, height=64, n_views=4, fx=90.0)
    g = sc.to(dev)
    imgs = torch.rand(4, 64, 96, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    out = []
    for fuse in (False, True):
        cfg = trainer.TrainConfig(max_steps=300, sh_degree_interval=1, fuse_adam=fuse, use_scale_regularization=True, scale_reg_every=4)
        mc = MCMCConfig(cap_max=2000, refine_start_iter=10_000, noise_lr=0.0, opacity_reg=0.01, scale_reg=0.01)
        tr = MCMCTrainer(g.params, g.viewmats, g.Ks, imgs, 96, 64, cfg, mc)
        for i in range(10):
            tr.step(i % 4)
        torch.cuda.synchronize()
        out.append(({k: tr.model.p(k)[: tr.model.n].clone() for k in trainer.GROUPS}, {k: tr.model.state(k, "m").clone() for k in trainer.GROUPS}))
    for k in trainer.GROUPS:
        assert rel_err(out[1][0][k], out[0][0][k]) < 2e-3 and rel_err(out[1][1][k], out[0][1][k]) < 2e-3, k
    # the regularisers reach Gaussians no view sees: their opacities and scales move identically in both runs
    assert not torch.equal(out[1][0]["opacities"], g.params["opacities"].reshape(out[1][0]["opacities"].shape))


def test_mcmc_regulariser_gradients_match_autograd(dev):
    """loss += opacity_reg * mean(sigmoid(o)) + scale_reg * mean(exp(s))  (gsplat simple_trainer, mcmc preset):
    the kernel ADDS exactly that gradient to what the backward left, with the 1/N and 1/(3N) of the two means."""
    ops = _ops()
    n = 777
    g = torch.Generator().manual_seed(5)
    o = (torch.randn(n, generator=g) * 2).double().requires_grad_(True)
    sl = (torch.randn(n, 3, generator=g) - 3).double().requires_grad_(True)
    (0.01 * torch.sigmoid(o).mean() + 0.02 * torch.exp(sl).mean()).backward()
    vo = torch.zeros(n, device=dev)
    vs = torch.zeros(n, 3, device=dev)
    o_d, sl_d = o.detach().float().to(dev), sl.detach().float().to(dev).contiguous()      # (named: a raw pointer keeps nothing alive)
    args = (n, ops._p(o_d), ops._p(sl_d), 0.01, 0.02, ops._p(vo), ops._p(vs), ops._stream(dev))
    ops._lib.call("mi3dgs_mcmc_regularise", *args)
    assert rel_err(vo.cpu(), o.grad) < 1e-5 and rel_err(vs.cpu(), sl.grad) < 1e-5
    ops._lib.call("mi3dgs_mcmc_regularise", *args)                      # accumulates (the backward's gradient is in there first)
    assert rel_err(vo.cpu(), 2 * o.grad) < 1e-5 and rel_err(vs.cpu(), 2 * sl.grad) < 1e-5


def test_no_chained_kernel_gave_up_waiting(dev):
    """The device-wide scan, the onesweep radix passes and the fused tile emit wait on one another with
    bounded spins; a wait that runs out leaves a bit in a sticky device word (and Trainer.refine raises on
    it).  After everything above ran in this process the word must be clean."""
    from mi3dgs import _lib
    ops = _ops()
    keys = torch.randint(0, 2 ** 31 - 1, (3_000_000,), device=dev, dtype=torch.int32)
    vals = torch.arange(3_000_000, device=dev, dtype=torch.int32)
    ops.sort_pairs_u32(keys, vals, 32)
    assert bool((keys[1:] >= keys[:-1]).all())
    assert _lib.async_errors(reset=False) == 0
