#!/bin/bash
out=gpurun_out/r03z; mkdir -p $out
for sk in 524288 196608; do
MI3DGS_OS_SMALL_KEYS=$sk MI3DGS_PROFILE_STEPS=20000:20200 timeout -k 10 200 python tools/train_wolf.py --steps 22000 --model splatfacto > $out/wolf_$sk.txt 2>&1; echo "== small_keys $sk"; grep -E "profile of|isect|depth|trained in" $out/wolf_$sk.txt | head -8 | cut -c1-160
done
