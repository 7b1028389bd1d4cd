"""Same-box timing of the loss kernels: the product library (32 x 32-tile kernels) against the experiments library with
MI3DGS_LOSS_STREAM=1 (row-streaming kernels, csrc/loss.hip), same inputs, results compared.
    python tools/loss_bench.py [H W] ..."""
import json
import os
import sys

os.environ["MI3DGS_LOSS_STREAM"] = "1"         # read by the experiments library only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
import torch                                    # noqa: E402


def main():
    from mi3dgs import _lib, ops
    dev = torch.device("cuda:0")
    sizes = [(1080, 1920), (720, 960), (800, 800)]
    if len(sys.argv) > 2:
        sizes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    st = ops._stream(dev)
    out = []
    for H, W in sizes:
        g = torch.Generator().manual_seed(1)
        a = torch.rand(1, H, W, 3, generator=g).to(dev)
        b = (a + 0.1 * torch.randn(1, H, W, 3, generator=g).to(dev)).clamp(0, 1)
        res = {}
        for name, call in (("stream", _lib.exp_call), ("tiles", _lib.call)):
            dm = [torch.empty_like(a) for _ in range(3)]
            sums = torch.zeros(2, device=dev)
            v = torch.empty_like(a)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tf, tb = [], []
            for rep in range(25):
                sums.zero_()
                e[0].record()
                call("mi3dgs_loss_fwd", 1, H, W, ops._p(a), ops._p(b), ops._p(dm[0]), ops._p(dm[1]), ops._p(dm[2]), ops._p(sums), st)
                e[1].record()
                call("mi3dgs_loss_bwd", 1, H, W, ops._p(a), ops._p(b), ops._p(dm[0]), ops._p(dm[1]), ops._p(dm[2]), 0.2, 1.0, ops._p(v), st)
                e[2].record()
                e[2].synchronize()
                if rep >= 5:
                    tf.append(e[0].elapsed_time(e[1]) * 1e3)
                    tb.append(e[1].elapsed_time(e[2]) * 1e3)
            res[name] = dict(fwd_us=sorted(tf)[len(tf) // 2], bwd_us=sorted(tb)[len(tb) // 2], sums=sums.tolist(), v=v, dm=dm)
        s, t = res["stream"], res["tiles"]
        rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())      # noqa: E731
        out.append(dict(H=H, W=W, stream_fwd_us=round(s["fwd_us"], 1), stream_bwd_us=round(s["bwd_us"], 1), tiles_fwd_us=round(t["fwd_us"], 1),
                        tiles_bwd_us=round(t["bwd_us"], 1), sums_stream=s["sums"], sums_tiles=t["sums"], v_rel_diff=rel(s["v"], t["v"]),
                        dm_rel_diff=[rel(x, y) for x, y in zip(s["dm"], t["dm"])]))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
