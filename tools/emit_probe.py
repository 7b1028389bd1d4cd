"""Phase stamps of the wave-granular tile emit on the bench workload (probe build of the library only):
     MI3DGS_LIB=$PWD/pipeline-pointcloud_amd/mi3dgs/libmi3dgs_stamps.so python tools/emit_probe.py"""
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
assert hasattr(L, "mi3dgs_debug_read_we_stamps"), "needs the probe build (-DMI3DGS_OS_STAMPS)"
buf = np.zeros((4096, 2, 8), dtype=np.uint64)
L.mi3dgs_debug_read_we_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
nb = int((buf[:, 0, 0] > 0).sum())
st = buf[:nb].astype(np.int64)
t0 = st[:, :, 0].min()
names = ["start", "gathered", "counted", "base known", "stores issued", "stores done"]
print(f"blocks {nb}; wave 0 / wave 15 of every block; us since the first block started")
for w in (0, 1):
    rel = (st[:, w, :6] - t0) / 100.0
    print(f" wave {'0' if w == 0 else '15'}: phase time (min / median / max over blocks)")
    for i, nm in enumerate(names):
        print(f"   {nm:14s} {rel[:, i].min():8.2f} {np.median(rel[:, i]):8.2f} {rel[:, i].max():8.2f}")
    d = np.diff(rel, axis=1)
    print("   durations (median / p90 / max):")
    for i in range(5):
        print(f"   {names[i]:14s} -> {names[i+1]:14s} {np.median(d[:, i]):7.2f} {np.percentile(d[:, i], 90):7.2f} {d[:, i].max():7.2f}")
    print("   rows per wave median / max:", int(np.median(st[:, w, 6])), int(st[:, w, 6].max()), " keys per wave median / max:", int(np.median(st[:, w, 7])), int(st[:, w, 7].max()))
rel0 = (st[:, 0, :6] - t0) / 100.0
idx = list(range(0, nb, max(1, nb // 16)))
print("by block index:", idx)
print(" start      ", [round(float(rel0[i, 0]), 1) for i in idx])
print(" chain wait ", [round(float(rel0[i, 3] - rel0[i, 2]), 1) for i in idx])
print(" done       ", [round(float(rel0[i, 5]), 1) for i in idx])
