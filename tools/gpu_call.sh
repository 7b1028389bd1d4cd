#!/bin/bash
out=gpurun_out/r03as; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 2 $out/suite.txt | cut -c1-300
for i in 1 2; do timeout -k 10 250 python tools/train_wolf.py --steps 30000 --model splatfacto > $out/wolf_$i.txt 2>&1; grep -E "eval:|trained in" $out/wolf_$i.txt | tr '\n' ' ' | cut -c1-200; echo; done
bash tools/profile_round.sh r03 final4
