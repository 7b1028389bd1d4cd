#!/bin/bash
out=gpurun_out/r03v; mkdir -p $out
for seg in True False True False; do
  timeout -k 10 250 python tools/train_wolf.py --steps 30000 --model splatfacto --mi3dgs.raster-segments $seg > $out/wolf_$seg.txt 2>&1; echo "segments=$seg: $(grep -E 'eval:|trained in' $out/wolf_$seg.txt | tr '\n' ' ' | cut -c1-220)"
done
