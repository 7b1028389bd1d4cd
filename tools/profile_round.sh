#!/bin/bash
# One gpurun call that refreshes everything under profiles/ for a round:
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_round.sh r01 final2'
# 1. bench.py (default flags, CPU baseline included)        -> gpurun_out/<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command   -> kernel_stats.{csv,txt}
# 3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs -> pmc_traffic.{json,txt}
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}_${2:-final}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[profile] bench" && timeout -k 10 420 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 $R/tools/show_bench.py $OUT/bench.json > $OUT/bench.txt; head -3 $OUT/bench.txt
echo "[profile] kernel trace" && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-profile > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
S=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
cp "$S" $OUT/kernel_stats.csv && python3 $R/tools/kstats.py $OUT/kernel_stats.csv 30 > $OUT/kernel_stats.txt; head -5 $OUT/kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  echo "[profile] pmc $c" && timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stage-profile > $OUT/pmc_$c.log 2>&1 || { tail -5 $OUT/pmc_$c.log; exit 1; }
done
F=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summary.py $F $W $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt; head -6 $OUT/pmc_traffic.txt
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
echo "[profile] done: $OUT"
