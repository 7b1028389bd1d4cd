#!/bin/bash
out=gpurun_out/r03u; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 4 $out/suite.txt | cut -c1-400
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -n 2 $out/smoke.txt | cut -c1-300
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1,8p | cut -c1-400
for sc in cube lego 6m; do timeout -k 10 300 python bench.py --scene $sc --no-cpu-baseline > $out/bench_$sc.json 2> $out/bench_$sc.err; python tools/show_bench.py $out/bench_$sc.json 2>/dev/null | sed -n 1p | cut -c1-200; done
