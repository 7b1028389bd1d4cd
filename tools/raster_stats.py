"""Debug: live-pair statistics of rasterize_bwd on the bench workload (needs `make STATS=1`)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
import torch
import bench
from mi3dgs import _lib

class A: scene="garden"; n=None; views=2; sync_isect=False; rehearse=False
sc, tr, V = bench.build_workload(A, 0, torch.device("cuda:0"))
h = _lib.lib()
buf = (ctypes.c_ulonglong * 8)()
h.mi3dgs_debug_raster_stats(buf, 1)
tr.step(0); torch.cuda.synchronize()
h.mi3dgs_debug_raster_stats(buf, 0)
I = int(tr.last["binning"]["n_isect"].item())
v = list(buf)
print("I", I, "visits", v[0], "live visits", v[1], "live lanes", v[2], "flushed", v[3], "staged", v[4])
print("visits/I %.2f  live-visit frac %.3f  lanes per live visit %.1f  flushed/staged %.3f  staged/I %.3f" % (
    v[0] / I, v[1] / max(v[0], 1), v[2] / max(v[1], 1), v[3] / max(v[4], 1), v[4] / I))
