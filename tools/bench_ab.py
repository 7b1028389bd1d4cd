"""bench.py with the library's A/B switches set first:  python tools/bench_ab.py emit=0 raster=1 -- [bench.py args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pipeline-pointcloud_amd"))
from mi3dgs import _lib
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
for a in args[:cut]:
    k, v = a.split("=")
    getattr(_lib.lib(), {"emit": "mi3dgs_debug_set_emit_mode", "raster": "mi3dgs_debug_set_raster_mode", "sort": "mi3dgs_debug_set_sort_mode"}[k])(int(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + args[cut + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
