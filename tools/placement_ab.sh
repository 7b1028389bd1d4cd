#!/bin/bash
mkdir -p gpurun_out/r2k
for i in 1 2 3; do
  MI3DGS_FLAT_MODEL=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-placement-tuning > gpurun_out/r2k/flat_$i.json 2> gpurun_out/r2k/flat_$i.err
  python - <<PY
import json; d=json.load(open("gpurun_out/r2k/flat_$i.json")); print("flat  no-tune", round(d["value"],1), "it/s  bwd_adam", round(d["stages"]["project_bwd_adam"]["us_per_launch"],1))
PY
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-placement-tuning > gpurun_out/r2k/sep_$i.json 2> gpurun_out/r2k/sep_$i.err
  python - <<PY
import json; d=json.load(open("gpurun_out/r2k/sep_$i.json")); print("separate no-tune", round(d["value"],1), "it/s  bwd_adam", round(d["stages"]["project_bwd_adam"]["us_per_launch"],1))
PY
done
