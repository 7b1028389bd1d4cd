#!/bin/bash
out=gpurun_out/r03c; mkdir -p $out
# 1. the race mechanism, made deterministic: wave 0 of os_hist_kernel<DROP> held back by ~8 x 127 x 64 cycles after the barrier
for v in r2_wave0late fixed_wave0late; do
  timeout -k 10 120 python tools/binning_stress.py --iters 40 --lib tools/ab/libmi3dgs_$v.so > $out/stress_$v.json 2>/dev/null; cut -c1-900 $out/stress_$v.json
done
# 2. rasterize_bwd: round-2 kernel / 128-slot batches at 3 waves per SIMD / at 4 (128 VGPRs, spills)
timeout -k 10 200 python tools/raster_ab.py --libs tools/ab/libmi3dgs_r2.so tools/ab/libmi3dgs_occ3.so tools/ab/libmi3dgs_occ4.so > $out/raster_ab_garden.json 2>/dev/null; cat $out/raster_ab_garden.json
timeout -k 10 200 python tools/raster_ab.py --scene lego --libs tools/ab/libmi3dgs_r2.so tools/ab/libmi3dgs_occ3.so tools/ab/libmi3dgs_occ4.so > $out/raster_ab_lego.json 2>/dev/null; grep -E "bwd_us_median|lib" $out/raster_ab_lego.json
# 3. project_bwd_adam: first-generation phase stagger (experiments build)
for us in 0 6 12 17 25 35; do
  MI3DGS_LIB=pipeline-pointcloud_amd/mi3dgs/libmi3dgs_exp.so MI3DGS_BWD_STAGGER_US=$us timeout -k 10 120 python bench.py --steps 30 --no-cpu-baseline > $out/bench_stagger_$us.json 2>/dev/null
  python - <<PY
import json
b=json.load(open("$out/bench_stagger_$us.json"))
print("stagger $us us:", round(b["value"],1), "it/s; project_bwd_adam", round(b["stages"]["project_bwd_adam"]["us_per_launch"],1), "us; copy", b["roofline"].get("copy_GBps_this_box"), "mix", b["roofline"].get("mix_5r4w_GBps_this_box"))
PY
done
# 4. new / changed tests
timeout -k 10 300 python -m pytest tests/test_gpu_dist.py tests/test_gpu_configs.py -m gpu -x -q > $out/tests.txt 2>&1; tail -n 4 $out/tests.txt
# 5. training
run() { name=$1; shift; echo "== $*" > $out/$name.txt; timeout -k 10 300 "$@" 2>&1 | grep -v "amdgpu.ids" >> $out/$name.txt; grep -E "eval: psnr" $out/$name.txt | cut -c1-120; }
export MI3DGS_MCMC_LOG=1 MI3DGS_EVAL_DETAIL=1
for i in 1 2 3 4; do run wolf_mcmc300k_$i python tools/train_wolf.py --steps 30000 --model splatfacto-mcmc --max-gaussians 300000; done
run synth_ns_mcmc python tools/train_synthetic.py --steps 30000 --mode ns-train --strategy mcmc --max-gaussians 300000
run wolf_st_mcmc python tools/train_wolf.py --steps 30000 --model splatfacto-mcmc --mode simple_trainer --max_gaussians 300000
run wolf_st_default python tools/train_wolf.py --steps 30000 --model splatfacto --mode simple_trainer
