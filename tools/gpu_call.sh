#!/bin/bash
out=gpurun_out/r03al; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 2 $out/suite.txt | cut -c1-300
MI3DGS_PROFILE_STEPS=20000:20200 timeout -k 10 200 python tools/train_wolf.py --steps 22000 --model splatfacto > $out/wolf_profile.txt 2>&1; grep -E "profile of|rasterize|eval:|trained in" $out/wolf_profile.txt | head -6 | cut -c1-160
