#!/bin/bash
out=gpurun_out/r03e; mkdir -p $out
for v in r2_wave0late fixed_wave0late; do
  timeout -k 10 120 python tools/binning_stress.py --iters 30 --lib tools/ab/libmi3dgs_$v.so > $out/stress_$v.json 2>/dev/null; cut -c1-1500 $out/stress_$v.json
done
L=pipeline-pointcloud_amd/mi3dgs/libmi3dgs.so
timeout -k 10 200 python tools/raster_ab.py --libs tools/ab/libmi3dgs_r2.so $L > $out/raster_ab_garden.json 2>/dev/null; grep -E "bwd_us_median|rel_diff|\"lib\"" $out/raster_ab_garden.json
timeout -k 10 200 python tools/raster_ab.py --scene lego --libs tools/ab/libmi3dgs_r2.so $L > $out/raster_ab_lego.json 2>/dev/null; grep -E "bwd_us_median|rel_diff|\"lib\"" $out/raster_ab_lego.json
timeout -k 10 200 python tools/raster_ab.py --scene wolf --absgrad --libs tools/ab/libmi3dgs_r2.so $L > $out/raster_ab_wolf_absgrad.json 2>/dev/null; grep -E "bwd_us_median|rel_diff|\"lib\"" $out/raster_ab_wolf_absgrad.json
timeout -k 10 200 python tools/raster_ab.py --scene garden --absgrad --libs tools/ab/libmi3dgs_r2.so $L > $out/raster_ab_garden_absgrad.json 2>/dev/null; grep -E "bwd_us_median|rel_diff|\"lib\"" $out/raster_ab_garden_absgrad.json
# three-term transport A/B (experiments library, modes 1 / 4 / 3 = product 2-term, 3-term, all-f32 reduce-scatter)
E=pipeline-pointcloud_amd/mi3dgs/libmi3dgs_exp.so
timeout -k 10 200 python tools/raster_ab.py --libs $E $E $E --modes 3 1 4 > $out/raster_terms_garden.json 2>/dev/null; grep -E "bwd_us_median|rel_diff|\"mode\"" $out/raster_terms_garden.json
timeout -k 10 200 python tools/raster_ab.py --scene lego --libs $E $E $E --modes 3 1 4 > $out/raster_terms_lego.json 2>/dev/null; grep -E "bwd_us_median|rel_diff|\"mode\"" $out/raster_terms_lego.json
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $out/suite.txt 2>&1; tail -n 6 $out/suite.txt | cut -c1-600
timeout -k 10 250 python bench.py > $out/bench.json 2> $out/bench.err; python tools/show_bench.py $out/bench.json | head -24
run() { name=$1; shift; echo "== $*" > $out/$name.txt; timeout -k 10 300 "$@" 2>&1 | grep -v "amdgpu.ids" >> $out/$name.txt; grep -E "eval: psnr|trained in" $out/$name.txt | cut -c1-120; }
export MI3DGS_MCMC_LOG=1 MI3DGS_EVAL_DETAIL=1
run wolf_splatfacto python tools/train_wolf.py --steps 30000 --model splatfacto
run synth_st_mcmc_noreg python tools/train_synthetic.py --steps 15000 --mode simple_trainer --strategy mcmc --max_gaussians 300000 --opacity_reg 0 --scale_reg 0
run synth_st_mcmc_noopareg python tools/train_synthetic.py --steps 15000 --mode simple_trainer --strategy mcmc --max_gaussians 300000 --opacity_reg 0
