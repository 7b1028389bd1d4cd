"""world_size-2 `gloo` tests of the multi-GPU host path: the same code runs
over RCCL on the GPUs.  The kernels themselves cannot run here, so these cover what the
collectives must guarantee: identical gradients, identical densify statistics and therefore
identical densify decisions on every replica, the view sharding, and bench.py's barrier +
max-over-ranks timing."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "pipeline-pointcloud_amd")):
            sys.path.insert(0, p)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        from mi3dgs import parallel
        from oracle import gs_oracle as O
        ctx = parallel.init_from_env(backend="gloo")
        assert ctx.active and ctx.rank == rank and ctx.world == world
        out = {}
        # 1. gradient mean over ranks
        g = torch.Generator().manual_seed(100 + rank)
        grads = [torch.randn(50, w, generator=g) for w in (3, 4, 3, 1, 3, 45)]
        ref = [t.clone() for t in grads]
        parallel.allreduce_mean_(grads, ctx)
        gather = [[torch.zeros_like(t) for _ in range(world)] for t in ref]
        for t, lst in zip(ref, gather):
            dist.all_gather(lst, t)
        out["grad_mean_ok"] = all(torch.allclose(a, torch.stack(l).mean(0), atol=1e-6) for a, l in zip(grads, gather))
        # 2. densify statistics -> identical decisions
        n = 200
        gs = torch.Generator().manual_seed(7)                      # same parameters on every replica
        scales = (torch.rand(n, 3, generator=gs) * 0.05).double()
        opac = torch.rand(n, generator=gs).double()
        gr = torch.Generator().manual_seed(200 + rank)             # different views -> different stats
        stats = dict(grad2d=torch.rand(n, generator=gr) * 4e-4, count=torch.randint(0, 3, (n,), generator=gr).float(),
                     radii=torch.rand(n, generator=gr))
        local = {k: v.clone() for k, v in stats.items()}
        parallel.allreduce_stats_(stats, ctx)
        allg = {k: [torch.zeros_like(v) for _ in range(world)] for k, v in local.items()}
        for k in local:
            dist.all_gather(allg[k], local[k])
        out["stats_ok"] = (torch.allclose(stats["grad2d"], torch.stack(allg["grad2d"]).sum(0)) and
                           torch.equal(stats["count"], torch.stack(allg["count"]).sum(0)) and
                           torch.equal(stats["radii"], torch.stack(allg["radii"]).max(0).values))
        d, s, p = O.strategy_masks(dict(grad2d=stats["grad2d"].double(), count=stats["count"].double()), scales, opac, 600)
        code = (d.long() + 2 * s.long() + 4 * p.long())
        codes = [torch.zeros_like(code) for _ in range(world)]
        dist.all_gather(codes, code)
        out["decisions_identical"] = all(torch.equal(codes[0], c) for c in codes)
        out["n_grow"] = int((d | s).sum())
        # 3. view sharding: a pass over the views touches each exactly once across ranks
        V = 10
        mine = [parallel.views_for_step(st, V, ctx) for st in range(V // world)]
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        out["views_ok"] = sorted(v for l in allv for v in l) == list(range(V))
        # 4. bench.py timing rule: barrier, then MAX over ranks
        dist.barrier()
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["max_time"] = float(t)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, out))
    except Exception as e:          # surface the failure in the parent
        import traceback
        q.put((rank, {"error": traceback.format_exc() + repr(e)}))


@pytest.mark.timeout(180)
def test_two_rank_gloo_collectives():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
    for r in range(world):
        assert "error" not in res[r], res[r].get("error")
        assert res[r]["grad_mean_ok"] and res[r]["stats_ok"] and res[r]["decisions_identical"] and res[r]["views_ok"]
        assert res[r]["max_time"] == float(world)
        assert res[r]["n_grow"] > 0


def test_batch_scaled_config_follows_the_sqrt_rule():
    from mi3dgs import parallel, trainer
    c = trainer.TrainConfig()
    s = parallel.batch_scaled_config(c, 4)
    assert abs(s.lr_means - 2 * c.lr_means) < 1e-12 and abs(s.adam_eps - c.adam_eps / 2) < 1e-24
    assert abs(s.adam_beta1 - 0.6) < 1e-9 and abs(s.adam_beta2 - 0.996) < 1e-9
    assert s.max_steps == c.max_steps // 4 and s.refine_every == 25 and s.reset_every == 750
    assert parallel.batch_scaled_config(c, 1) is c


def test_single_process_context_is_inactive():
    from mi3dgs import parallel
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    ctx = parallel.init_from_env()
    assert not ctx.active and ctx.world == 1
    t = [torch.ones(3)]
    parallel.allreduce_mean_(t, ctx)
    assert torch.equal(t[0], torch.ones(3))


# ------------------------------------------------------------- sharded optimiser
def _torch_adam_step(params, grads, exp_avg, exp_avg_sq, lrs, step, beta1=0.9, beta2=0.999, eps=1e-15, numel=None, grad_scale=1.0):
    """Stand-in for ops.adam_step on CPU tensors (torch.optim.Adam arithmetic); the HIP kernel itself is checked
    against torch.optim.Adam on the GPU (tests/test_gpu_configs.py)."""
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avg, exp_avg_sq)):
        n = p.numel() if numel is None else int(numel[i])
        pf, gf, mf, vf = p.reshape(-1)[:n], g.reshape(-1)[:n], m.reshape(-1)[:n], v.reshape(-1)[:n]
        if grad_scale != 1.0:
            gf = gf * grad_scale              # what the HIP kernel does on the way in (mi3dgs_adam_step, grad_scale)
        mf.mul_(beta1).add_(gf, alpha=1 - beta1)
        vf.mul_(beta2).addcmul_(gf, gf, value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
        pf.addcdiv_(mf, vf.sqrt() / bc2 ** 0.5 + eps, value=-lrs[i] / bc1)


def _sharded_worker(rank, world, port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "pipeline-pointcloud_amd")):
            sys.path.insert(0, p)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        from mi3dgs import ops, parallel, scenes, trainer
        ops.adam_step = _torch_adam_step
        ctx = parallel.init_from_env(backend="gloo")
        sc = scenes.make_cube(n=301, seed=5, width=32, height=32, n_views=2, fx=30.0)     # 301: slices cut through groups
        imgs = torch.zeros(2, 32, 32, 3)
        out = {}
        trs = {}
        for shard in (True, False):
            cfg = trainer.TrainConfig(capacity=333, fuse_adam=False)
            trs[shard] = parallel.DataParallelTrainer(sc.params, sc.viewmats, sc.Ks, imgs, 32, 32, cfg, ctx=ctx, shard_optimizer=shard)
        A, B = trs[True], trs[False]
        out["flat"] = A.model.flat is not None and B.model.flat is None and A.model.capacity % (4 * world) == 0
        pieces = A._slice_pieces()
        tot = sum(trainer.WIDTHS) * A.model.capacity
        out["slice_elems"] = sum(c for _, _, c in pieces)
        n = A.model.n
        # the work is cut by LIVE rows -- with n well below the capacity every rank still owns its share of every
        # group (rows differ by at most one alignment unit of 4 * world), and only the live rows travel
        cap0, n0 = A.model.capacity, A.model.n
        A.model.n = n0 // 2
        rows = [c // w for (_, _, c), w in zip(A._slice_pieces(), trainer.WIDTHS)]
        allrows = [None] * world
        dist.all_gather_object(allrows, rows)
        per_rank = [r_[0] if r_ else 0 for r_ in allrows]
        out["balanced"] = (all(len(r_) == 6 and len(set(r_)) == 1 for r_ in allrows if r_) and sum(per_rank) == n0 // 2
                           and max(per_rank) - min(per_rank) <= 4 * world)
        out["bytes_follow_live_rows"] = A.xgmi_bytes_per_step() < 0.6 * int(2 * (world - 1) / world * 4 * sum(trainer.WIDTHS) * cap0)
        A.model.n = n0
        for step in range(3):
            g = torch.Generator().manual_seed(1000 * step + rank)             # every rank saw a different view
            for tr in (A, B):
                g2 = torch.Generator().manual_seed(1000 * step + rank)
                for gi, name in enumerate(trainer.GROUPS):
                    tr.model.grads[name].zero_()
                    tr.model.grads[name][:n] = torch.randn(n, trainer.WIDTHS[gi], generator=g2) * 10.0 ** (-gi)
                tr._optimizer_step(n)
                tr.step_count += 1
        # (a ring all-reduce sums in an order that depends on where an element sits in its buffer, so flat and per-group
        # reductions agree bit for bit only at world 2; replicas of ONE method must always be identical)
        same = torch.equal if world == 2 else (lambda a, b: torch.allclose(a, b, rtol=1e-5, atol=1e-7))
        out["same_as_dense"] = all(same(A.model.p(k), B.model.p(k)) for k in trainer.GROUPS)
        out["moved"] = not torch.equal(A.model.p("means"), sc.params["means"])
        out["in_sync"] = A.replicas_in_sync() and B.replicas_in_sync()
        stale = not all(torch.equal(A.model.state(k, "m"), B.model.state(k, "m")) for k in trainer.GROUPS)
        A._sync_optimizer_state()
        out["moments_after_sync"] = all(same(A.model.state(k, s_), B.model.state(k, s_)) for k in trainer.GROUPS for s_ in ("m", "v"))
        out["moments_were_sharded"] = stale
        out["bytes"] = (A.xgmi_bytes_per_step(), B.xgmi_bytes_per_step())
        # the elements of all ranks' slices tile the live part of the flat buffer exactly once
        allp = [None] * world
        dist.all_gather_object(allp, pieces)
        cover = torch.zeros(tot, dtype=torch.int32)
        for pl in allp:
            for _, a, c in pl:
                cover[a: a + c] += 1
        live = torch.zeros(tot, dtype=torch.int32)
        off = 0
        for w in trainer.WIDTHS:
            live[off: off + w * n] = 1
            off += w * A.model.capacity
        out["tiling"] = bool(torch.equal(cover, live))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, out))
    except Exception as e:
        import traceback
        q.put((rank, {"error": traceback.format_exc() + repr(e)}))


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_equals_dense_all_reduce(world):
    """reduce_scatter -> Adam on a 1/G slice -> all_gather of the parameters gives, bit for bit, what the dense
    gradient mean + the full Adam step on every rank gives; replicas stay identical; the moments each rank holds
    are current only on its slice until _sync_optimizer_state() (called at refine)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=200) for _ in range(world))
    for p in procs:
        p.join(30)
    for r in range(world):
        assert "error" not in res[r], res[r].get("error")
        o = res[r]
        assert o["flat"] and o["same_as_dense"] and o["moved"] and o["in_sync"] and o["tiling"], o
        assert o["balanced"] and o["bytes_follow_live_rows"], o
        assert o["moments_were_sharded"] and o["moments_after_sync"], o
        assert o["bytes"][0] > 0 and o["bytes"][1] > 0


def _probe_rank(rank, world, port, a):
    """What the launcher test runs in every spawned process: rendezvous over gloo with the launcher's arguments."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    with open(os.path.join(a["result_dir"], f"rank{rank}.txt"), "w") as f:
        f.write(f"{world} {float(t)} {a['strategy']} {a['steps_scaler']}")
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_simple_trainer_launcher_spawns_one_process_per_visible_gpu(tmp_path, monkeypatch):
    """main_simple_trainer is gsplat's launcher: one process per visible GPU, own rendezvous (main.py:1318-1347 only
    sets the env).  A fake device count stands in for the GPUs this container does not have."""
    for p in (ROOT, os.path.join(ROOT, "pipeline-pointcloud_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mi3dgs import cli
    monkeypatch.setenv("MI3DGS_FAKE_DEVICE_COUNT", "3")
    argv = ["mcmc", "--max_steps", "30000", "--result-dir", str(tmp_path), "--data_factor", "1", "--steps_scaler", str(1 / 3),
            "--disable_viewer", "--packed", "--batch-size", "1", "--data-dir", str(tmp_path)]
    assert cli.main_simple_trainer(argv, rank_fn=_probe_rank) == 0
    for r in range(3):
        world, total, strat, scaler = open(tmp_path / f"rank{r}.txt").read().split()
        assert int(world) == 3 and float(total) == 6.0 and strat == "mcmc" and abs(float(scaler) - 1 / 3) < 1e-9
    # forced single-GPU run: the reference's 1/G step scaling is undone
    monkeypatch.setenv("MI3DGS_SINGLE_GPU", "1")
    seen = {}
    cli.main_simple_trainer(argv, rank_fn=lambda rank, world, port, a: seen.update(world=world, scaler=a["steps_scaler"]))
    assert seen["world"] == 1 and abs(seen["scaler"] - 1.0) < 1e-9


def test_view_order_is_shared_and_covers_every_view():
    from mi3dgs import cli
    V, world = 10, 4
    orders = [cli.ViewOrder(V, seed=3) for _ in range(world)]
    seen = []
    for step in range(5):                                    # 20 slots = two epochs
        views = [orders[r](step * world + r) for r in range(world)]
        seen += views
    assert sorted(seen[:V]) == list(range(V)) and sorted(seen[V:2 * V]) == list(range(V))
    assert seen[:V] != seen[V:2 * V]                         # reshuffled between epochs
    assert [cli.ViewOrder(V, seed=3)(s) for s in range(20)] == seen          # every rank derives the same order
