#!/bin/bash
out=gpurun_out/r03d; mkdir -p $out
# 1. the race mechanism made deterministic (delay pinned in front of the LDS reads this time)
for v in r2_wave0late fixed_wave0late; do
  timeout -k 10 120 python tools/binning_stress.py --iters 30 --lib tools/ab/libmi3dgs_$v.so > $out/stress_$v.json 2>/dev/null; cut -c1-1200 $out/stress_$v.json
done
# 2. full suite (new: RCCL branches, principled crop exclusion, 6M crop)
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 6 $out/suite.txt | cut -c1-400
# 3. reached fractions
timeout -k 10 200 python tools/reached_fraction.py > $out/reached.txt 2>/dev/null; cat $out/reached.txt | cut -c1-400
# 4. tile sort through the onesweep passes (32-bit keys) instead of the classic 16-bit ones
for mk in default 67108864; do
  if [ $mk = default ]; then unset MI3DGS_OS_MAX_KEYS; else export MI3DGS_OS_MAX_KEYS=$mk; fi
  timeout -k 10 120 python bench.py --steps 30 --no-cpu-baseline > $out/bench_osmax_$mk.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("$out/bench_osmax_$mk.json")); s=b["stages"]
print("OS_MAX_KEYS $mk:", round(b["value"],1), "it/s", round(b["render_fps"],1), "fps;", {k: round(v["ms_per_step"]*1e3,1) for k,v in s.items() if "isect" in k or k in ("tile_emit","tile_offsets")})
PY
done
unset MI3DGS_OS_MAX_KEYS
# 5. is small-scene training host-bound?  kernel time per step against the wall clock
MI3DGS_PROFILE_STEPS=20000:20200 timeout -k 10 200 python tools/train_wolf.py --steps 22000 --model splatfacto > $out/wolf_profile.txt 2>&1; grep -E "profile of|ms/step|step 2(0|1)[0-9]01/|eval:" $out/wolf_profile.txt | cut -c1-160
# 6. training on the coherent synthetic scene
run() { name=$1; shift; echo "== $*" > $out/$name.txt; timeout -k 10 300 "$@" 2>&1 | grep -v "amdgpu.ids" >> $out/$name.txt; grep -E "eval: psnr" $out/$name.txt | cut -c1-120; }
export MI3DGS_MCMC_LOG=1 MI3DGS_EVAL_DETAIL=1
run synth_st_default python tools/train_synthetic.py --steps 30000 --mode simple_trainer --strategy default
run synth_ns_splatfacto python tools/train_synthetic.py --steps 30000 --mode ns-train --strategy default
run synth_st_mcmc python tools/train_synthetic.py --steps 30000 --mode simple_trainer --strategy mcmc --max_gaussians 300000
run synth_ns_mcmc python tools/train_synthetic.py --steps 30000 --mode ns-train --strategy mcmc --max-gaussians 300000
run synth_ns_big python tools/train_synthetic.py --steps 30000 --mode ns-train --strategy default --model-big
