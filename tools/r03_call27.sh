#!/bin/bash
out=gpurun_out/r03aa; mkdir -p $out
L=pipeline-pointcloud_amd/mi3dgs/libmi3dgs.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "segments or backward or absgrad" > $out/seg_tests.txt 2>&1; tail -n 2 $out/seg_tests.txt | cut -c1-300
run() { timeout -k 10 300 python tools/raster_ab.py "$@" --libs $L $L --seg 0 1 > $out/seg_var.json 2>$out/seg_var.err; python - <<PY
import json
d=json.load(open("$out/seg_var.json"))
print("$*", "n_isect", d["n_isect"], " | ".join(f"bwd {r['bwd_us_median']:.1f} fwd {r['fwd_us_median']:.1f} diff {r['rel_diff_vs_first']:.1e}" for r in d["results"]))
PY
}
run --scene garden
run --scene wolf
run --scene lego
timeout -k 10 250 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1p | cut -c1-200; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 3,8p
