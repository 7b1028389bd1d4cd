import torch, time
dev = torch.device("cuda:0")
N = 2_000_000
for W in (16, 8):
    tab = torch.randn(N, W, device=dev)
    idx = torch.randperm(N, device=dev)[:1_330_000]
    out = torch.empty(idx.numel(), W, device=dev)
    for _ in range(3): torch.index_select(tab, 0, idx, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    junk = torch.empty(300_000_000, dtype=torch.uint8, device=dev)
    ts = []
    for _ in range(5):
        junk.zero_()
        e0.record(); torch.index_select(tab, 0, idx, out=out); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"random gather of 1.33 M rows of {W * 4} B from a {N * W * 4 / 1e6:.0f} MB table (cold caches): {min(ts):.1f} us")
    ts = []
    for _ in range(5):
        e0.record(); torch.index_select(tab, 0, idx, out=out); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"  warm: {min(ts):.1f} us")
