#!/bin/bash
out=gpurun_out/r03ay; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "culled_groups" > $out/t.txt 2>&1; tail -n 2 $out/t.txt | cut -c1-300
for w in 0 256 512 1024 2048 0 256 512 1024 2048; do
  timeout -k 10 250 python bench.py --no-cpu-baseline --no-stage-profile --culled-adam-waves $w > $out/bench.json 2> $out/bench.err; echo "waves $w: $(python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1p | cut -c1-70)"
done
