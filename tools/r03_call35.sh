#!/bin/bash
out=gpurun_out/r03ag; mkdir -p $out
for v in "" "--no-spatial-sort" "" "--no-spatial-sort"; do
  timeout -k 10 250 python bench.py --no-cpu-baseline $v > $out/bench.json 2> $out/bench.err; echo "== sort: ${v:-on}"; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1p | cut -c1-120; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 3,9p
done
