#!/bin/bash
out=gpurun_out/r03aj; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "loss" > $out/loss_tests.txt 2>&1; tail -n 2 $out/loss_tests.txt | cut -c1-300
timeout -k 10 200 python tools/loss_bench.py > $out/loss_bench.json 2> $out/loss_bench.err; grep -E "\"H\"|tiles_fwd_us|tiles_bwd_us" $out/loss_bench.json | paste - - - | cut -c1-200
