"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files into per-kernel
HBM bytes per launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
counters are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x (so both the raw
and the doubled figure are kept); WRITE_SIZE is exact for 16-B streaming stores and float atomics.

usage: python tools/pmc_summary.py <fetch.csv> <write.csv> [out.json]
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("mfma_raster::", "").replace("void ", "")
    # bench.py's profiler tags: the fused single-camera backward is "project_bwd_adam"
    n = n.replace("project_bwd1_kernel<true, true>", "project_bwd_adam_probe").replace("project_bwd1_kernel<true, false>", "project_bwd_adam")
    n = n.replace("project_bwd1_kernel<false, false>", "project_bwd")
    n = n.replace("project_bwd1_kernel<true>", "project_bwd_adam").replace("project_bwd1_kernel<false>", "project_bwd")
    m = re.match(r"rasterize_bwd_mm_kernel<(\w+), (\w+),", n)
    if m:       # <HAS_BG, ABSGRAD, ...>: the `ns-train splatfacto` preset (bench.py's default) runs the ABSGRAD instance
        return "rasterize_bwd_mm" if m.group(2) == "true" else "rasterize_bwd_mm_plain"
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"<.*", "", n)
    return n.replace("_kernel", "")


def load(path, counter):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return per


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for key in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) + sum(w.get(k, [0])))):
        name, grid = key
        fv, wv = f.get(key, [0.0]), w.get(key, [0.0])
        fr = 1024.0 * sum(fv) / len(fv)
        wr = 1024.0 * sum(wv) / len(wv)
        tag = name if name not in out else f"{name}@grid{grid}"
        out[tag] = dict(grid=grid, launches=len(fv), fetch_bytes_raw=fr, fetch_bytes_x2=2 * fr, write_bytes=wr,
                        hbm_bytes_per_launch=2 * fr + wr)
    for k, v in list(out.items())[:24]:
        print(f"{k:36s} grid {v['grid']:>10d}  fetch(raw) {v['fetch_bytes_raw']/1e6:9.1f} MB  write {v['write_bytes']/1e6:9.1f} MB"
              f"  hbm(2F+W) {v['hbm_bytes_per_launch']/1e6:9.1f} MB")
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
