#!/bin/bash
out=gpurun_out/r03s; mkdir -p $out
for h in first first first churn churn first bench bench first churn; do
  timeout -k 10 150 python tools/pbwd_placement.py --history $h 2> $out/err.txt | tee -a $out/placement.jsonl || { tail -n 3 $out/err.txt; exit 1; }
done
