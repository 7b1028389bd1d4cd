#!/bin/bash
out=gpurun_out/r03r; mkdir -p $out
MI3DGS_LIB=pipeline-pointcloud_amd/mi3dgs/libmi3dgs_exp.so MI3DGS_LOSS_STREAM=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "loss" > $out/loss_tests.txt 2>&1; tail -n 3 $out/loss_tests.txt | cut -c1-400
timeout -k 10 200 python tools/loss_bench.py > $out/loss_bench.json 2> $out/loss_bench.err; grep -v "sums\|\[\|\]\|{\|}" $out/loss_bench.json | paste - - - - - - - | cut -c1-250; tail -n 3 $out/loss_bench.err
