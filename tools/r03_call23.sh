#!/bin/bash
out=gpurun_out/r03w; mkdir -p $out
E=pipeline-pointcloud_amd/mi3dgs/libmi3dgs_exp.so
run() { timeout -k 10 300 python tools/raster_ab.py "$@" --libs $E $E --modes 21 22 --seg 1 1 > $out/x.json 2>$out/x.err; python - <<PY
import json
d=json.load(open("$out/x.json"))
print("$*", "n_isect", d["n_isect"], " | ".join(f"mode {r['mode']} items {r['seg_items']} bwd {r['bwd_us_median']:.1f}" for r in d["results"]))
PY
}
run --scene wolf
run --scene wolf --absgrad
run --scene wolf --wolf-size 640 480
run --scene wolf --wolf-size 1920 1080
run --scene lego
run --scene lego --absgrad
