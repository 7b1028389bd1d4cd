#!/bin/bash
# Round 3: the reference's actual job length (MAX_STEPS = 30000, config.json:22) through the shims, on datasets whose every
# pixel shows content (tools/train_*.py --backdrop, the default).
# usage: bash tools/r03_long_runs.sh <outdir> [steps] [extra args for every run, e.g. --backdrop 0]
out=${1:-gpurun_out/r03_train}; steps=${2:-30000}; shift 2
mkdir -p $out
run() { # name, command...
  name=$1; shift
  echo "== $*" > $out/$name.txt
  timeout -k 10 400 "$@" 2>&1 | grep -v "amdgpu.ids" >> $out/$name.txt
  tail -n 1 $out/$name.txt | cut -c1-200
}
export MI3DGS_MCMC_LOG=1 MI3DGS_EVAL_DETAIL=1
run wolf_splatfacto      python tools/train_wolf.py --steps $steps --model splatfacto "$@"
run wolf_big             python tools/train_wolf.py --steps $steps --model splatfacto-big "$@"
run wolf_mcmc            python tools/train_wolf.py --steps $steps --model splatfacto-mcmc "$@"
run synth_st_default     python tools/train_synthetic.py --steps $steps --mode simple_trainer --strategy default "$@"
run synth_ns_splatfacto  python tools/train_synthetic.py --steps $steps --mode ns-train --strategy default "$@"
run synth_st_mcmc        python tools/train_synthetic.py --steps $steps --mode simple_trainer --strategy mcmc "$@"
run known_st_default     python tools/train_synthetic.py --steps $steps --mode simple_trainer --strategy default --gt 50000 --points 50000 --views 120 --seed-noise 0.002 "$@"
