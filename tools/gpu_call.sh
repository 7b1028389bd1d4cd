#!/bin/bash
out=gpurun_out/r03az; mkdir -p $out
for sc in cube lego garden; do
  timeout -k 10 280 python tools/binning_stress.py --iters 1500 --scene $sc > $out/stress_$sc.txt 2>&1; echo "== $sc"; tail -n 3 $out/stress_$sc.txt | cut -c1-300
done
