#!/bin/bash
out=gpurun_out/r03ae; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 2 $out/suite.txt | cut -c1-300
for i in 1 2; do timeout -k 10 250 python tools/train_wolf.py --steps 30000 --model splatfacto > $out/wolf_$i.txt 2>&1; grep -E "eval:|trained in" $out/wolf_$i.txt | tr '\n' ' ' | cut -c1-200; echo; done
timeout -k 10 250 python tools/train_wolf.py --steps 30000 --model splatfacto-mcmc > $out/wolf_mcmc.txt 2>&1; grep -E "eval:|trained in" $out/wolf_mcmc.txt | tr '\n' ' ' | cut -c1-200; echo
timeout -k 10 250 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1p | cut -c1-200; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 3,5p
for sc in cube lego; do timeout -k 10 250 python bench.py --scene $sc --no-cpu-baseline > $out/bench_$sc.json 2> $out/bench_$sc.err; python tools/show_bench.py $out/bench_$sc.json 2>/dev/null | sed -n 1p | cut -c1-120; done
