#!/bin/bash
out=gpurun_out/r03p; mkdir -p $out
L=pipeline-pointcloud_amd/mi3dgs/libmi3dgs.so
A=tools/ab
for sz in "1920 1080" "1280 720" "640 480"; do
  timeout -k 10 300 python tools/raster_ab.py --scene wolf --wolf-size $sz --libs $L $L $A/libmi3dgs_seg256.so --seg 0 1 1 > $out/seg_var.json 2>$out/seg_var.err; echo "== wolf $sz"; python - <<PY
import json
d=json.load(open("$out/seg_var.json"))
print("n_isect", d["n_isect"])
for r in d["results"]:
    print(r["lib"].split("/")[-1], r["seg_items"], round(r["bwd_us_median"],1), round(r["fwd_us_median"],1), "%.1e"%r["rel_diff_vs_first"])
PY
done
timeout -k 10 300 python tools/raster_ab.py --scene lego --libs $L $L $A/libmi3dgs_seg256.so --seg 0 1 1 > $out/seg_var.json 2>$out/seg_var.err; echo "== lego"; python - <<PY
import json
d=json.load(open("$out/seg_var.json"))
for r in d["results"]:
    print(r["lib"].split("/")[-1], r["seg_items"], round(r["bwd_us_median"],1), round(r["fwd_us_median"],1), "%.1e"%r["rel_diff_vs_first"])
PY
