#!/bin/bash
# second batch of tools/r04_precision_ab.sh: the first showed a launch-to-launch spread of the SAME code (0.3 - 0.55 dB) larger than
# any difference between the variants, so every variant gets more launches and the means are compared
out=${1:-gpurun_out/r04_precision}; steps=${2:-30000}
mkdir -p $out
run() { name=$1; shift; echo "== $*" > $out/$name.txt; timeout -k 10 500 "$@" 2>&1 | grep -v "amdgpu.ids" >> $out/$name.txt; tail -n 1 $out/$name.txt | cut -c1-300; }
for r in c d; do
run ns_product_$r   python3 tools/train_wolf.py --steps $steps --model splatfacto
run ns_f32_$r       python3 tools/train_wolf.py --steps $steps --model splatfacto --raster-mode 3
run st_product_$r   python3 tools/train_wolf.py --steps $steps --mode simple_trainer --model default
run st_f32_$r       python3 tools/train_wolf.py --steps $steps --mode simple_trainer --model default --raster-mode 3
run st_three_term_$r python3 tools/train_wolf.py --steps $steps --mode simple_trainer --model default --raster-mode 4
done
run ns_f32_e       python3 tools/train_wolf.py --steps $steps --model splatfacto --raster-mode 3
