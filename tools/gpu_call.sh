#!/bin/bash
out=gpurun_out/r03ar; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "second_stream or culled_groups or fused_adam" > $out/t.txt 2>&1; tail -n 4 $out/t.txt | cut -c1-600
for v in off after_binning after_raster_fwd off after_binning after_raster_fwd; do
  timeout -k 10 250 python bench.py --no-cpu-baseline --no-stage-profile --overlap-adam $v > $out/bench.json 2> $out/bench.err; echo "overlap $v: $(python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1p | cut -c1-90)"
done
timeout -k 10 250 python bench.py --scene 6m --no-cpu-baseline --no-stage-profile --overlap-adam off > $out/bench.json 2> $out/bench.err; echo "6m off: $(python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1p | cut -c1-90)"
timeout -k 10 250 python bench.py --scene 6m --no-cpu-baseline --no-stage-profile --overlap-adam after_binning > $out/bench.json 2> $out/bench.err; echo "6m after_binning: $(python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1p | cut -c1-90)"
