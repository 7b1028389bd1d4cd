#!/bin/bash
# SQ counter passes of the rasteriser kernels only, for one raster mode:  bash tools/sq_raster.sh <mode> <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
MODE=${1:-1}
OUT=$R/gpurun_out/sq_raster_${2:-m$MODE}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MI3DGS_RASTER_MODE=$MODE
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_LDS_UNALIGNED_STALL" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-profile > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; }
  F=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && cp "$F" $OUT/pmc_sq_$i.csv
  rm -rf $OUT/p$i
done
python3 $R/tools/pmc_sq.py $OUT/pmc_sq_*.csv --k=rasterize_fwd,rasterize_bwd,rasterize_bwd_mm > $OUT/pmc_sq.txt 2>&1
cat $OUT/pmc_sq.txt
