#!/bin/bash
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/suite_final.txt 2>&1; tail -n 2 gpurun_out/suite_final.txt | cut -c1-300
bash tools/profile_round.sh r03 final3
