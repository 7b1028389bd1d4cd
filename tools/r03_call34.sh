#!/bin/bash
out=gpurun_out/r03af; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "segments or backward or absgrad or clears" > $out/seg_tests.txt 2>&1; tail -n 2 $out/seg_tests.txt | cut -c1-300
for i in 1 2; do timeout -k 10 250 python tools/train_wolf.py --steps 30000 --model splatfacto > $out/wolf_$i.txt 2>&1; grep -E "eval:|trained in" $out/wolf_$i.txt | tr '\n' ' ' | cut -c1-200; echo; done
MI3DGS_PROFILE_STEPS=1000:1200 timeout -k 10 100 python tools/train_wolf.py --steps 1300 --model splatfacto > $out/wolf_early.txt 2>&1; grep -E "profile of|ms/step" $out/wolf_early.txt | head -6 | cut -c1-160
