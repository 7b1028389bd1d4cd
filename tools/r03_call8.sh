#!/bin/bash
out=gpurun_out/r03h; mkdir -p $out
L=pipeline-pointcloud_amd/mi3dgs/libmi3dgs.so
E=pipeline-pointcloud_amd/mi3dgs/libmi3dgs_exp.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "segments or backward or absgrad" > $out/seg_tests.txt 2>&1; tail -n 5 $out/seg_tests.txt | cut -c1-800
# segments off / on, same library, same lists
for sc in wolf garden lego; do for ag in "" "--absgrad"; do
  timeout -k 10 200 python tools/raster_ab.py --scene $sc $ag --libs $L $L --seg 0 1 > $out/raster_seg_${sc}${ag}.json 2>$out/raster_seg_${sc}${ag}.err; echo "== seg $sc $ag"; grep -E "bwd_us_median|fwd_us_median|rel_diff|seg_items" $out/raster_seg_${sc}${ag}.json | paste - - - - | cut -c1-200
done; done
# the two shapes of the product backward (experiments library: modes 21 = DEEP, 22 = WIDE)
for sc in garden lego wolf; do for ag in "" "--absgrad"; do
  timeout -k 10 200 python tools/raster_ab.py --scene $sc $ag --libs $E $E --modes 21 22 > $out/raster_shape_${sc}${ag}.json 2>/dev/null; echo "== shape $sc $ag"; grep -E "bwd_us_median|rel_diff|\"mode\"" $out/raster_shape_${sc}${ag}.json | paste - - - | cut -c1-170
done; done
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 6 $out/suite.txt | cut -c1-600
timeout -k 10 250 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python tools/show_bench.py $out/bench.json | head -8
MI3DGS_PROFILE_STEPS=20000:20200 timeout -k 10 200 python tools/train_wolf.py --steps 22000 --model splatfacto > $out/wolf_profile.txt 2>&1; grep -E "profile of|ms/step|eval:|trained in|rasterize" $out/wolf_profile.txt | head -14 | cut -c1-160
