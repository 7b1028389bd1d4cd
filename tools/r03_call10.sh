#!/bin/bash
out=gpurun_out/r03j; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 8 $out/suite.txt | cut -c1-700
timeout -k 10 250 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; python tools/show_bench.py $out/bench.json | head -9
MI3DGS_PROFILE_STEPS=20000:20200 timeout -k 10 200 python tools/train_wolf.py --steps 22000 --model splatfacto > $out/wolf_profile.txt 2>&1; grep -E "profile of|ms/step|eval:|trained in" $out/wolf_profile.txt | head -8 | cut -c1-160
