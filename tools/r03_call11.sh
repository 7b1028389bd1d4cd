#!/bin/bash
out=gpurun_out/r03k; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "loss or training_recovers" > $out/loss_tests.txt 2>&1; tail -n 4 $out/loss_tests.txt | cut -c1-400
timeout -k 10 200 python tools/loss_bench.py > $out/loss_bench.json 2> $out/loss_bench.err; grep -v "sums\|\[\|\]\|{\|}" $out/loss_bench.json | paste - - - - - - - | cut -c1-250; tail -n 3 $out/loss_bench.err
