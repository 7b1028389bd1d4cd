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
