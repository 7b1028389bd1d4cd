#!/bin/bash
out=gpurun_out/r03q; mkdir -p $out
L=pipeline-pointcloud_amd/mi3dgs/libmi3dgs.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "segments or backward or absgrad" > $out/seg_tests.txt 2>&1; tail -n 2 $out/seg_tests.txt | cut -c1-300
run() { timeout -k 10 300 python tools/raster_ab.py "$@" --libs $L $L --seg 0 1 > $out/seg_var.json 2>$out/seg_var.err; python - <<PY
import json
d=json.load(open("$out/seg_var.json"))
print("$*", "n_isect", d["n_isect"], " | ".join(f"{r['seg_items']} bwd {r['bwd_us_median']:.1f} fwd {r['fwd_us_median']:.1f} diff {r['rel_diff_vs_first']:.1e}" for r in d["results"]))
PY
}
run --scene wolf --wolf-size 1920 1080
run --scene wolf --wolf-size 1280 720
run --scene wolf
run --scene wolf --absgrad
run --scene wolf --wolf-size 640 480
run --scene lego
run --scene garden
MI3DGS_PROFILE_STEPS=20000:20200 timeout -k 10 200 python tools/train_wolf.py --steps 22000 --model splatfacto > $out/wolf_profile.txt 2>&1; grep -E "profile of|rasterize_bwd|rasterize_fwd|eval:" $out/wolf_profile.txt | head -5 | cut -c1-160
