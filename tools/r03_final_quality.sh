#!/bin/bash
# End of round 3: one 30 000-step run per strategy through the shims on the wolf + backdrop dataset, final library and defaults.
out=gpurun_out/r03_quality; mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 "$@" > $out/$name.txt 2>&1; echo "$name: $(grep -E "eval:|trained in|\[wolf\] result" $out/$name.txt | tr '\n' ' ' | cut -c1-220)"; }
run wolf_splatfacto python tools/train_wolf.py --steps 30000 --model splatfacto
run wolf_big        python tools/train_wolf.py --steps 30000 --model splatfacto-big
run wolf_mcmc_300k  python tools/train_wolf.py --steps 30000 --model splatfacto-mcmc --max-gaussians 300000
run wolf_st_default python tools/train_wolf.py --steps 30000 --mode simple_trainer --model default
run wolf_st_mcmc    python tools/train_wolf.py --steps 30000 --mode simple_trainer --model mcmc
