"""Every C-ABI entry that takes per-Gaussian or per-intersection arrays, with EVERY input, output and workspace ending at the end
of a device allocation of its own (docs/FINDINGS_r03.md 3b).

The kernels launch whole 256-thread blocks over N Gaussians and several of them use a "load first, clamp the index" idiom; an idle
lane whose clamped index is computed wrongly reads behind its array.  Inside a caching allocator's segment that read hits
somebody else's bytes and is discarded; where the array ends with its segment the page behind is not mapped and the device
faults (round 3: `project_bwd_adam`, found by an unlucky order of tests).  Here nothing is left to luck: a tensor's last byte
lies less than 16 bytes (the alignment the library requires of a base address) in front of the end of a 12 MiB allocation, which
torch's allocator serves as a segment of its own (requests of 10 MiB and more are not carved out of shared blocks).  N runs
through 1, 63, 65 and 77 modulo 256: a lone Gaussian in the last block, a wave short by one, a wave with one lane, an odd tail.

Each call runs twice, ordinarily placed and end-placed; the results must agree (bit for bit where the kernel is deterministic).
"""
import math

import pytest
import torch

from helpers import assert_clean, rel_err, small_scene

pytestmark = pytest.mark.gpu

SEG = 12 << 20


class EndPlacer:
    """Stands in for the `torch` module inside mi3dgs.ops: every device tensor ops allocates (outputs, workspaces) ends at the
    end of its own allocation; everything else is torch's."""

    def __init__(self):
        self.keep = []

    def __getattr__(self, name):
        return getattr(torch, name)

    def place(self, shape, dtype, device, fill=None):
        if isinstance(shape, int):
            shape = (shape,)
        shape = tuple(int(x) for x in shape)
        item = torch.empty(0, dtype=dtype).element_size()
        nbytes = item * math.prod(shape)
        total = max(SEG, (nbytes + 64 + (2 << 20) - 1) // (2 << 20) * (2 << 20))
        buf = torch.empty(total, dtype=torch.uint8, device=device)
        start = (total - nbytes) // 16 * 16
        assert total - (start + nbytes) < 16
        v = buf[start:start + nbytes].view(dtype).view(shape)
        if fill is not None:
            v.fill_(fill)
        self.keep.append(buf)
        return v

    def at_end(self, t):
        v = self.place(t.shape, t.dtype, t.device)
        v.copy_(t)
        return v

    @staticmethod
    def _dev(device):
        return device is not None and torch.device(device).type == "cuda"

    def empty(self, *size, dtype=torch.float32, device=None, **kw):
        if not self._dev(device):
            return torch.empty(*size, dtype=dtype, device=device, **kw)
        return self.place(size[0] if len(size) == 1 and not isinstance(size[0], int) else size, dtype, device)

    def zeros(self, *size, dtype=torch.float32, device=None, **kw):
        if not self._dev(device):
            return torch.zeros(*size, dtype=dtype, device=device, **kw)
        return self.place(size[0] if len(size) == 1 and not isinstance(size[0], int) else size, dtype, device, fill=0)

    def empty_like(self, t, **kw):
        return self.place(t.shape, kw.get("dtype", t.dtype), t.device) if t.is_cuda else torch.empty_like(t, **kw)

    def zeros_like(self, t, **kw):
        return self.place(t.shape, kw.get("dtype", t.dtype), t.device, fill=0) if t.is_cuda else torch.zeros_like(t, **kw)


class placed:
    """with placed(end) as P: inside, mi3dgs.ops allocates through an EndPlacer (end=True) or through torch (end=False);
    P.put(t) places an input accordingly."""

    def __init__(self, end):
        self.end = end
        self.placer = EndPlacer()

    def put(self, t):
        return self.placer.at_end(t.contiguous()) if self.end else t.contiguous().clone()

    def __enter__(self):
        from mi3dgs import ops
        self._ops, self._torch, self._ws = ops, ops.torch, dict(ops._WS)
        ops._WS.clear()                       # cached workspaces are re-made under this placement
        if self.end:
            ops.torch = self.placer
        return self

    def __exit__(self, *exc):
        self._ops.torch = self._torch
        self._ops._WS.clear()
        self._ops._WS.update(self._ws)
        torch.cuda.synchronize()
        return False


NS = [1024 + r for r in (1, 63, 65, 77)]
W, H = 96, 64
LRS = (1.6e-4, 1e-3, 5e-3, 5e-2, 2.5e-3, 1.25e-4)


def _scene(n, dev):
    sc = small_scene(n=n, seed=60 + n % 7, big=True, width=W, height=H, n_views=1, fx=70.0)
    sc.params["means"][: n // 5, 2] += 30.0          # a fifth of them behind the far side: culled rows among the visible ones
    return sc.to(dev)


def _pipeline(g, n, end, absgrad):
    """projection -> binning (capacity = the exact count: the lists end with their buffers) -> both rasterisers -> both
    projection backwards -> Adam, with every tensor placed as `end` says.  Returns what each stage produced."""
    from mi3dgs import ops, trainer
    dev = g.params["means"].device
    out = {}
    fl = ops.FLAG_LOG_SCALES | ops.FLAG_LOGIT_OPAC
    with placed(end) as P:
        prm = {k: P.put(g.params[k].float()) for k in trainer.GROUPS}
        vm, K = P.put(g.viewmats[:1]), P.put(g.Ks[:1])
        keys = ops.torch.empty(1, n, dtype=torch.int32, device=dev)
        radii, splats = ops.project_fwd(prm["means"], prm["quats"], prm["scales"], prm["opacities"], vm, K, W, H, sh0=prm["sh0"],
                                        shN=prm["shN"], sh_degree=3, flags=fl, depth_keys=keys)
        out["radii"], out["splats"] = radii.clone(), splats.clone()
        exact = ops.bin_tiles(radii, splats, W, H, 16, tight=True, radii_in_records=True)            # two-phase, sized exactly
        I = int(exact["n_isect"].item())
        out["I"] = I
        keys2 = P.put(keys)
        b = ops.bin_tiles(radii, splats, W, H, 16, max_isect=I, tight=True, fused=True, depth_keys=keys2, radii_in_records=True,
                          want_tile_keys=False)                                                      # fused, capacity == count
        for k in ("flatten_ids", "isect_offsets"):
            out["exact_" + k], out["fused_" + k] = exact[k].clone(), b[k].clone()
        out["fused_n"] = int(b["n_isect"].item())
        bg = P.put(torch.tensor([[0.2, 0.3, 0.4]], device=dev))
        ro = {}
        ws = ops.raster_seg_workspace(b, 1, dev, ro)
        render, alphas, last = ops.rasterize_fwd(splats, b, W, H, 16, bg, ro, seg_ws=ws)
        out["render"], out["alphas"], out["last"] = render.clone(), alphas.clone(), last.clone()
        gen = torch.Generator().manual_seed(n)
        vr = P.put((torch.rand(1, H, W, 3, generator=gen) - 0.5).to(dev))
        va = P.put((torch.rand(1, H, W, 1, generator=gen) - 0.5).to(dev))
        v_splats = ops.torch.zeros(1, n, ops.GRAD_STRIDE, device=dev)
        ops.rasterize_bwd(splats, b, W, H, alphas, last, vr, va, 16, bg, absgrad, v_splats, render=render, seg_ws=ws)
        out["v_splats"] = v_splats.clone()
        # unfused projection backward + statistics, then Adam over the six groups
        stats = {k: ops.torch.zeros(n, device=dev) for k in ("grad2d", "count", "radii")}
        grads = {"v_" + k: ops.torch.zeros(prm[k].shape, device=dev) for k in trainer.GROUPS}
        ops.project_bwd(prm["means"], prm["quats"], prm["scales"], prm["opacities"], vm, K, W, H, radii, splats, P.put(v_splats),
                        sh0=prm["sh0"], shN=prm["shN"], color_mode=ops.COLOR_SH, sh_degree=3, flags=fl, out=grads, stats=stats,
                        stat_use_abs=absgrad)
        for k in trainer.GROUPS:
            out["g_" + k] = grads["v_" + k].clone()
        for k in stats:
            out["stat_" + k] = stats[k].clone()
        pa = [P.put(prm[k].reshape(n, -1)) for k in trainer.GROUPS]
        m1 = [ops.torch.zeros(x.shape, device=dev) for x in pa]
        m2 = [ops.torch.zeros(x.shape, device=dev) for x in pa]
        ops.adam_step(pa, [grads["v_" + k].reshape(n, -1) for k in trainer.GROUPS], m1, m2, LRS, 1)
        for k, x in zip(trainer.GROUPS, pa):
            out["adam_" + k] = x.clone()
        # the fused backward + Adam, all Gaussians and visible / culled groups apart (the single-GPU step's three launches)
        for tag, flags in (("fused", 0), ("split", ops.FLAG_ONLY_VISIBLE_GROUPS)):
            pf = [P.put(prm[k].reshape(n, -1)) for k in trainer.GROUPS]
            f1 = [ops.torch.zeros(x.shape, device=dev) for x in pf]
            f2 = [ops.torch.zeros(x.shape, device=dev) for x in pf]
            vs = P.put(v_splats)
            if flags:
                ops.adam_culled_groups(pf, f1, f2, LRS, 1, radii, n=n)
            ops.project_bwd_adam(pf, f1, f2, LRS, 1, vm, K, W, H, radii, splats, vs, n=n, sh_degree=3,
                                 flags=fl | ops.FLAG_CLEAR_VSPLATS | flags, stats={k: ops.torch.zeros(n, device=dev) for k in stats},
                                 stat_use_abs=absgrad)
            for k, x, y in zip(trainer.GROUPS, pf, f2):
                out[f"{tag}_{k}"], out[f"{tag}_v_{k}"] = x.clone(), y.clone()
            out[f"{tag}_cleared"] = vs.clone()
        assert_clean(ops, f"end-of-allocation pipeline n={n} end={end}")
    return out


@pytest.mark.parametrize("absgrad", [False, True])
@pytest.mark.parametrize("n", NS)
def test_step_kernels_read_and_write_nothing_behind_their_arrays(dev, n, absgrad):
    g = _scene(n, dev)
    a = _pipeline(g, n, False, absgrad)
    b = _pipeline(g, n, True, absgrad)
    assert a["I"] == b["I"] == b["fused_n"] > 2000 and int((a["radii"] > 0).all(-1).sum()) < n      # something is culled
    exact = ["radii", "splats", "exact_flatten_ids", "exact_isect_offsets", "fused_flatten_ids", "fused_isect_offsets", "render",
             "alphas", "last"]
    for k in exact:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(b["exact_flatten_ids"], b["fused_flatten_ids"][: b["I"]])
    # float atomics meet in another order from launch to launch: everything behind rasterize_bwd agrees to rounding
    for k in a:
        if k in exact or k in ("I", "fused_n"):
            continue
        assert bool(torch.isfinite(b[k]).all()), k
        assert rel_err(b[k], a[k]) < 2e-5, (k, rel_err(b[k], a[k]))
    # the fused launch and the visible / culled pair are the same update
    from mi3dgs import trainer
    for k in trainer.GROUPS:
        assert rel_err(b["split_" + k], b["fused_" + k]) < 2e-5 and rel_err(b["fused_" + k], b["adam_" + k]) < 2e-4, k
    assert float(b["fused_cleared"].abs().max()) == 0.0


@pytest.mark.parametrize("n", NS)
def test_refine_kernels_at_the_end_of_their_allocations(dev, n):
    """densify_decide -> scan -> densify_scatter (18 arrays in, 18 out, flags, offsets, map), reset_opacity and the MCMC kernels."""
    from mi3dgs import ops, trainer
    g = _scene(n, dev)
    gen = torch.Generator().manual_seed(n)
    st = {k: (torch.rand(n, generator=gen) * s).to(dev) for k, s in (("grad2d", 6e-4), ("count", 3.0), ("radii", 0.2))}
    res = []
    for end in (False, True):
        with placed(end) as P:
            src = {k: [P.put(g.params[k].float().reshape(n, -1)), P.put(torch.full((n, w), 1.0, device=dev)),
                       P.put(torch.full((n, w), 2.0, device=dev))] for k, w in zip(trainer.GROUPS, trainer.WIDTHS)}
            s3 = {k: P.put(v) for k, v in st.items()}
            flags = ops.torch.empty(n, dtype=torch.uint8, device=dev)
            counts = ops.torch.empty(n, dtype=torch.int32, device=dev)
            offs = ops.torch.empty(n, dtype=torch.int32, device=dev)
            total = ops.torch.zeros(1, dtype=torch.int32, device=dev)
            stream = ops._stream(dev)
            ops._lib.call("mi3dgs_densify_decide", n, ops._p(src["scales"][0]), ops._p(src["opacities"][0]), ops._p(s3["grad2d"]),
                          ops._p(s3["count"]), ops._p(s3["radii"]), 2e-4, 0.01, 0.05, 0.005, 0.1, 0.15, 1, 1, ops._p(flags), ops._p(counts), stream)
            ops.scan_exclusive_u32(counts, offs, total)
            new_n = int(total.item())
            dst = {k: [ops.torch.zeros(new_n, w, device=dev) for _ in range(3)] for k, w in zip(trainer.GROUPS, trainer.WIDTHS)}
            mapw = ops.torch.empty(new_n, dtype=torch.int32, device=dev)
            G = trainer.GROUPS
            ops._lib.call("mi3dgs_densify_scatter", n, new_n, ops._ptr_array([src[k][0] for k in G]), ops._ptr_array([src[k][1] for k in G]),
                          ops._ptr_array([src[k][2] for k in G]), ops._ptr_array([dst[k][0] for k in G]), ops._ptr_array([dst[k][1] for k in G]),
                          ops._ptr_array([dst[k][2] for k in G]), ops._p(flags), ops._p(offs), new_n, 1234, ops._p(mapw), stream)
            opa = dst["opacities"]
            ops._lib.call("mi3dgs_reset_opacity", new_n, ops._p(opa[0]), -4.0, ops._p(opa[1]), ops._p(opa[2]), stream)
            ops._lib.call("mi3dgs_mcmc_inject_noise", new_n, ops._p(dst["means"][0]), ops._p(dst["quats"][0]), ops._p(dst["scales"][0]),
                          ops._p(opa[0]), 0.5, 99, stream)
            vo, vs = ops.torch.zeros(new_n, device=dev), ops.torch.zeros(new_n, 3, device=dev)
            ops._lib.call("mi3dgs_mcmc_regularise", new_n, ops._p(opa[0]), ops._p(dst["scales"][0]), 0.01, 0.01, ops._p(vo), ops._p(vs), stream)
            assert_clean(ops, f"refine kernels n={n} end={end}")
            res.append(dict(flags=flags.clone(), offs=offs.clone(), new_n=new_n, vo=vo.clone(), vs=vs.clone(),
                            **{f"{k}{i}": dst[k][i].clone() for k in G for i in range(3)}))
    a, b = res
    assert a["new_n"] == b["new_n"] and b["new_n"] != n
    for k in a:
        if k != "new_n":
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("n", NS + [70_000 + 77])
def test_sort_scan_and_knn_at_the_end_of_their_allocations(dev, n):
    from mi3dgs import ops
    gen = torch.Generator().manual_seed(n)
    keys = torch.randint(0, 1 << 30, (n,), generator=gen, dtype=torch.int32).to(dev)
    pts = torch.rand(n, 3, generator=gen).to(dev)
    res = []
    for end in (False, True):
        with placed(end) as P:
            k, v = P.put(keys), P.put(torch.arange(n, dtype=torch.int32, device=dev))
            ops.sort_pairs_u32(k, v, 30)                                   # (70 077 keys: the passes of a sort in ONE launch)
            x = P.put((keys & 15))
            sc, tot = ops.torch.empty(n, dtype=torch.int32, device=dev), ops.torch.zeros(1, dtype=torch.int32, device=dev)
            ops.scan_exclusive_u32(x, sc, tot)
            d = ops.knn(P.put(pts), 3)
            assert_clean(ops, f"sort / scan / knn n={n} end={end}")
            res.append((k.clone(), v.clone(), sc.clone(), tot.clone(), d.clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)
    assert bool((res[1][0][1:] >= res[1][0][:-1]).all())
