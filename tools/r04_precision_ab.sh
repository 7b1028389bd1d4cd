#!/bin/bash
# does the 16-bit transport of rasterize_bwd's pixel sums show at the end of a 30 000-step job?  The same dataset
# (the reference's wolf.spz + backdrop, 60 views 960x720, held-out = every 8th), the same seed, through the shims:
#   ns-train splatfacto (absgrad):   product backward x2 (the launch-to-launch spread: float atomics), all-f32 reduce-scatter backward
#   simple_trainer default (plain):  product, all-f32, three bf16 terms
out=${1:-gpurun_out/r04_precision}; steps=${2:-30000}
mkdir -p $out
run() { name=$1; shift; echo "== $*" > $out/$name.txt; timeout -k 10 500 "$@" 2>&1 | grep -v "amdgpu.ids" >> $out/$name.txt; tail -n 1 $out/$name.txt | cut -c1-400; }
run ns_product_a   python3 tools/train_wolf.py --steps $steps --model splatfacto
run ns_product_b   python3 tools/train_wolf.py --steps $steps --model splatfacto
run ns_f32         python3 tools/train_wolf.py --steps $steps --model splatfacto --raster-mode 3
run st_product_a   python3 tools/train_wolf.py --steps $steps --mode simple_trainer --model default
run st_product_b   python3 tools/train_wolf.py --steps $steps --mode simple_trainer --model default
run st_f32         python3 tools/train_wolf.py --steps $steps --mode simple_trainer --model default --raster-mode 3
run st_three_term  python3 tools/train_wolf.py --steps $steps --mode simple_trainer --model default --raster-mode 4
