#!/bin/bash
out=gpurun_out/r03ax; mkdir -p $out
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1,2p | cut -c1-500
