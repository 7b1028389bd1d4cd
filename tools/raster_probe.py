"""Block timeline of rasterize_bwd on the bench workload (probe build of the library only):
     MI3DGS_LIB=$PWD/pipeline-pointcloud_amd/mi3dgs/libmi3dgs_stamps.so python tools/raster_probe.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
import torch
import bench
from mi3dgs import _lib

sys.argv = [sys.argv[0], "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-stage-profile"]
args = bench.parse()
dev = torch.device("cuda:0")
sc, tr, V = bench.build_workload(args, 0, dev)
for i in range(4):
    tr.step(i % V)
torch.cuda.synchronize()
L = _lib.lib()
assert hasattr(L, "mi3dgs_debug_read_rb_stamps"), "needs the probe build (-DMI3DGS_OS_STAMPS)"
buf = np.zeros((16384, 3), dtype=np.uint64)
L.mi3dgs_debug_read_rb_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
tw, th = (sc.width + 15) // 16, (sc.height + 15) // 16
nt = tw * th
st = buf[:nt].astype(np.int64)
worked = st[:, 1] > 0
t0 = st[:, 0].min()
s0 = (st[:, 0] - t0) / 100.0
e0 = np.where(worked, (st[:, 1] - t0) / 100.0, s0)
span = e0.max()
work = st[:, 2]
print(f"tiles {nt} ({int(worked.sum())} with work); kernel span {span:.1f} us; list length median {int(np.median(work[worked]))} max {int(work.max())}")
dur = (e0 - s0)[worked]
print(f"block duration us: median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f};  us per 1000 list entries: {np.median(dur / np.maximum(work[worked], 1) * 1000):.2f}")
bins = 24
edges = np.linspace(0, span, bins + 1)
act = [(np.minimum(e0, edges[i + 1]) - np.maximum(s0, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) for i in range(bins)]
print("blocks active on average per 1/24 of the span (768 = 3 per CU):")
print(" ", [int(round(a)) for a in act])
order = np.argsort(e0)
cw = np.cumsum(work[order]) / max(1, work.sum())
for f in (0.5, 0.9, 0.95, 0.99):
    print(f"  {int(f * 100)} % of the list entries are done at {e0[order][np.searchsorted(cw, f)]:.1f} us")
last = np.argsort(-e0)[:8]
print("last blocks to finish: (tile x, y, start us, end us, list length)", [(int(t % tw), int(t // tw), round(float(s0[t]), 1), round(float(e0[t]), 1), int(work[t])) for t in last])
print("start time of block index 0, 1000, ...:", [round(float(s0[i]), 1) for i in range(0, nt, 1000)])
