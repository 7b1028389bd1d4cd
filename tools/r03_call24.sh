#!/bin/bash
out=gpurun_out/r03x; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 2 $out/suite.txt | cut -c1-300
MI3DGS_PROFILE_STEPS=20000:20200 timeout -k 10 200 python tools/train_wolf.py --steps 22000 --model splatfacto > $out/wolf_profile.txt 2>&1; grep -E "profile of|ms/step|eval:|trained in" $out/wolf_profile.txt | head -16 | cut -c1-160
timeout -k 10 250 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 1p | cut -c1-200; python tools/show_bench.py $out/bench.json 2>/dev/null | sed -n 4,9p
timeout -k 10 250 python bench.py --scene lego --no-cpu-baseline > $out/bench_lego.json 2> $out/bench_lego.err; python tools/show_bench.py $out/bench_lego.json 2>/dev/null | sed -n 1p | cut -c1-200
