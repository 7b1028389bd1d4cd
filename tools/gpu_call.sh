#!/bin/bash
out=gpurun_out/r03am; mkdir -p $out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -n 2 $out/smoke.txt | cut -c1-300
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1,3p | cut -c1-600
