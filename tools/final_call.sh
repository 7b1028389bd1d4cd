#!/bin/bash
# end-of-round: rocprofv3 kernel trace of the bench command (the backward rasteriser changed after the committed stats)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-profile > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
S=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
cp "$S" $OUT/kernel_stats.csv && python3 $R/tools/kstats.py $OUT/kernel_stats.csv 30 > $OUT/kernel_stats.txt; head -6 $OUT/kernel_stats.txt
rm -rf $OUT/kt
