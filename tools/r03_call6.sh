#!/bin/bash
out=gpurun_out/r03f; mkdir -p $out
L=pipeline-pointcloud_amd/mi3dgs/libmi3dgs.so; A=tools/ab/libmi3dgs_bwdA.so; C=tools/ab/libmi3dgs_bwdC.so
for sc in garden lego wolf; do
  timeout -k 10 200 python tools/raster_ab.py --scene $sc --libs $L $A $C > $out/raster_shape_$sc.json 2>/dev/null; echo "== $sc"; grep -E "bwd_us_median|rel_diff" $out/raster_shape_$sc.json | paste - - | cut -c1-120
  timeout -k 10 200 python tools/raster_ab.py --scene $sc --absgrad --libs $L $A $C > $out/raster_shape_${sc}_absgrad.json 2>/dev/null; echo "== $sc absgrad"; grep -E "bwd_us_median|rel_diff" $out/raster_shape_${sc}_absgrad.json | paste - - | cut -c1-120
done
# three-term transport A/B (experiments library; modes 3 = all-f32 reduce-scatter, 1 = product two-term, 4 = three-term)
E=pipeline-pointcloud_amd/mi3dgs/libmi3dgs_exp.so
for sc in garden lego; do
  timeout -k 10 200 python tools/raster_ab.py --scene $sc --libs $E $E $E --modes 3 1 4 > $out/raster_terms_$sc.json 2>/dev/null; echo "== terms $sc"; grep -E "bwd_us_median|rel_diff|\"mode\"" $out/raster_terms_$sc.json | paste - - - | cut -c1-160
done
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 6 $out/suite.txt | cut -c1-600
timeout -k 10 250 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python tools/show_bench.py $out/bench.json | head -20
