#!/bin/bash
out=gpurun_out/r03av; mkdir -p $out
for v in "" tools/ab/libmi3dgs_adamscalar.so "" tools/ab/libmi3dgs_adamscalar.so; do
  for o in after_binning after_project; do
  MI3DGS_LIB=$v timeout -k 10 250 python bench.py --no-cpu-baseline --no-stage-profile --overlap-adam $o > $out/bench.json 2> $out/bench.err; echo "lib ${v:-product} $o: $(python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1p | cut -c1-90)"
  done
done
