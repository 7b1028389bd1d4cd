#!/bin/bash
out=gpurun_out/r03an; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_configs.py -x -q -k "segments" > $out/t.txt 2>&1; tail -n 4 $out/t.txt | cut -c1-400
