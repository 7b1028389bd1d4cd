#!/bin/bash
# usage: bash tools/run_synth.sh <outfile> <args to tools/train_synthetic.py ...>
out=$1; shift
mkdir -p $(dirname $out)
echo "== tools/train_synthetic.py $*" >> $out
timeout -k 10 500 python tools/train_synthetic.py "$@" 2>&1 | grep -v "amdgpu.ids" >> $out
