#!/bin/bash
out=gpurun_out/r03o; mkdir -p $out
L=pipeline-pointcloud_amd/mi3dgs/libmi3dgs.so
A=tools/ab
for sc in wolf garden; do
  timeout -k 10 300 python tools/raster_ab.py --scene $sc --libs $L $L $A/libmi3dgs_seg256.so $A/libmi3dgs_seg256w1024.so $A/libmi3dgs_seg512w1024.so $A/libmi3dgs_seg1024.so --seg 0 1 1 1 1 1 > $out/seg_var_$sc.json 2>$out/seg_var_$sc.err; echo "== $sc"; python - <<PY
import json
d=json.load(open("$out/seg_var_$sc.json"))
for r in d["results"]:
    print(r["lib"].split("/")[-1], r["seg_items"], round(r["bwd_us_median"],1), round(r["fwd_us_median"],1), "%.1e"%r["rel_diff_vs_first"])
PY
done
