#!/bin/bash
out=gpurun_out/r03i; mkdir -p $out
L=pipeline-pointcloud_amd/mi3dgs/libmi3dgs.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "segments or backward or absgrad" > $out/seg_tests.txt 2>&1; tail -n 3 $out/seg_tests.txt | cut -c1-400
for sc in wolf garden lego; do
  timeout -k 10 200 python tools/raster_ab.py --scene $sc --libs $L $L --seg 0 1 > $out/raster_seg_${sc}.json 2>$out/raster_seg_${sc}.err; echo "== seg $sc"; grep -E "bwd_us_median|fwd_us_median|rel_diff|seg_items" $out/raster_seg_${sc}.json | paste - - - - | cut -c1-200
done
timeout -k 10 400 python tests/diag_crop.py garden 0 880 560 160 96 > $out/diag_garden.txt 2>&1; cut -c1-260 $out/diag_garden.txt | grep -v Warn | head -70
