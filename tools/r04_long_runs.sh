#!/bin/bash
# Round 4: the reference's job length (MAX_STEPS = 30000, config.json:22) through the shims on the wolf dataset, every model name the
# shim accepts (profiles/r04_train_30000.txt).  usage: bash tools/r04_long_runs.sh <outdir> [steps]
out=${1:-gpurun_out/r04_train}; steps=${2:-30000}
mkdir -p $out
run() { name=$1; shift; echo "== $*" > $out/$name.txt; timeout -k 10 400 "$@" 2>&1 | grep -v "amdgpu.ids" >> $out/$name.txt; tail -n 1 $out/$name.txt | cut -c1-260; }
export MI3DGS_MCMC_LOG=1
run wolf_splatfacto      python3 tools/train_wolf.py --steps $steps --model splatfacto
run wolf_big             python3 tools/train_wolf.py --steps $steps --model splatfacto-big
run wolf_mcmc            python3 tools/train_wolf.py --steps $steps --model splatfacto-mcmc
run wolf_st_default      python3 tools/train_wolf.py --steps $steps --mode simple_trainer --model default
run wolf_st_mcmc         python3 tools/train_wolf.py --steps $steps --mode simple_trainer --model mcmc
