#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03ad; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/train_wolf.py --steps 6000 --model splatfacto > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
S=$(find $out/kt -name "*kernel_stats.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/kstats.py $S 40 > $out/kernel_stats_wolf.txt; cat $out/kernel_stats_wolf.txt | cut -c1-150
rm -rf $out/kt
