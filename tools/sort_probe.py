"""Time the stable u32 pair sort alone (tools only): python tools/sort_probe.py [n_keys] [nbits]
   MI3DGS_OS_NOLOOKBACK=1 gives the (wrong-result) time of the onesweep passes without their look-back."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "pipeline-pointcloud_amd"))
import torch
from mi3dgs import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_330_000
nbits = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
# depth-like keys: positive floats viewed as u32
depth = (0.5 + 20.0 * torch.rand(n, device=dev, generator=g)).float()
keys0 = depth.view(torch.int32).clone() if nbits == 32 else torch.randint(0, 1 << nbits, (n,), device=dev, dtype=torch.int32, generator=g)
vals0 = torch.arange(n, device=dev, dtype=torch.int32)
for rep in range(3):
    keys, vals = keys0.clone(), vals0.clone()
    ops.sort_pairs_u32(keys, vals, nbits)
torch.cuda.synchronize()
ok = bool((keys[1:].to(torch.int64) >= keys[:-1].to(torch.int64)).all()) if nbits < 32 else bool((keys.view(torch.float32)[1:] >= keys.view(torch.float32)[:-1]).all())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for rep in range(20):
    keys.copy_(keys0); vals.copy_(vals0)
    e0.record()
    ops.sort_pairs_u32(keys, vals, nbits)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"n={n} nbits={nbits} sorted={ok} median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f} us  env NOLOOKBACK={os.environ.get('MI3DGS_OS_NOLOOKBACK','')}")

# probe build (MI3DGS_LIB=.../libmi3dgs_stamps.so, -DMI3DGS_OS_STAMPS): phase stamps of the tiles of the LAST pass
from mi3dgs import _lib
L = _lib.lib()
if hasattr(L, "mi3dgs_debug_read_os_stamps"):
    import ctypes, numpy as np
    buf = np.zeros((1024, 8), dtype=np.uint64)
    torch.cuda.synchronize()
    L.mi3dgs_debug_read_os_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
    tile = 256 * (8 if n <= 512 * 1024 else 32)
    nt = min(1024, (n + tile - 1) // tile)
    st = buf[:nt].astype(np.int64)
    t0 = st[:, 0].min()
    rel = (st - t0) / 100.0          # us (100 MHz)
    names = ["start", "ranked", "barrier", "published", "lookback done", "lds scatter", "stores issued", "stores done"]
    print(f"tiles {nt}; per phase: time since the first tile started, us (min / median / max over tiles)")
    for i, nm in enumerate(names):
        c = rel[:, i]
        print(f"  {nm:16s} {c.min():7.2f} {np.median(c):7.2f} {c.max():7.2f}")
    d = np.diff(rel, axis=1)
    print("per-tile phase durations, us (median / max):")
    for i in range(7):
        print(f"  {names[i]:14s} -> {names[i+1]:14s} {np.median(d[:, i]):6.2f} {d[:, i].max():6.2f}")
    print("lookback duration by tile index (every 16th):", [round(float(x), 2) for x in d[::16, 3]])
    print("start time by tile index (every 16th):", [round(float(x), 2) for x in rel[::16, 0]])
