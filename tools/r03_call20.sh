#!/bin/bash
out=gpurun_out/r03t; mkdir -p $out
S=tools/ab/libmi3dgs_rbstamps.so
for sc in wolf lego garden; do
timeout -k 10 200 python tools/raster_ab.py --scene $sc --libs $S --seg 1 > $out/probe_$sc.json 2>$out/probe_$sc.err; python - <<PY
import json
d=json.load(open("$out/probe_$sc.json")); print("$sc", d["n_isect"], round(d["results"][0]["bwd_us_median"],1), d["probe"])
PY
done
