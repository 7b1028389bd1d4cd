#!/bin/bash
out=gpurun_out/r03ao; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "second_stream or morton or fused_adam" > $out/t.txt 2>&1; tail -n 4 $out/t.txt | cut -c1-500
for v in off after_project after_binning off after_project after_binning; do
  timeout -k 10 250 python bench.py --no-cpu-baseline --no-stage-profile --overlap-adam $v > $out/bench.json 2> $out/bench.err; echo "overlap $v: $(python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1p | cut -c1-90)"
done
