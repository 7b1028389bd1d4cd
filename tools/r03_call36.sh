#!/bin/bash
out=gpurun_out/r03ah; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 3 $out/suite.txt | cut -c1-400
for i in 1 2; do timeout -k 10 250 python tools/train_wolf.py --steps 30000 --model splatfacto > $out/wolf_$i.txt 2>&1; grep -E "eval:|trained in" $out/wolf_$i.txt | tr '\n' ' ' | cut -c1-200; echo; done
for sc in cube lego 6m; do timeout -k 10 300 python bench.py --scene $sc --no-cpu-baseline > $out/bench_$sc.json 2> $out/bench_$sc.err; python tools/show_bench.py $out/bench_$sc.json 2>/dev/null | sed -n 1p | cut -c1-120; done
timeout -k 10 300 python bench.py --scene 6m --no-cpu-baseline --no-spatial-sort > $out/bench_6m_nosort.json 2> $out/bench_6m_nosort.err; python tools/show_bench.py $out/bench_6m_nosort.json 2>/dev/null | sed -n 1p | cut -c1-120
