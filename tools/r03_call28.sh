#!/bin/bash
out=gpurun_out/r03ab; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "radix or binning or scan or chained" > $out/sort_tests.txt 2>&1; tail -n 2 $out/sort_tests.txt | cut -c1-300
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 2 $out/suite.txt | cut -c1-300
MI3DGS_PROFILE_STEPS=20000:20200 timeout -k 10 200 python tools/train_wolf.py --steps 22000 --model splatfacto > $out/wolf_profile.txt 2>&1; grep -E "profile of|depth|isect|eval:|trained in" $out/wolf_profile.txt | head -9 | cut -c1-160
for sc in cube lego; do timeout -k 10 250 python bench.py --scene $sc --no-cpu-baseline > $out/bench_$sc.json 2> $out/bench_$sc.err; python tools/show_bench.py $out/bench_$sc.json 2>/dev/null | sed -n 1p | cut -c1-120; python tools/show_bench.py $out/bench_$sc.json 2>/dev/null | grep -E "depth|isect"; done
