#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2u; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  export MI3DGS_EMIT_MODE=$m
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-profile > $OUT/log_$m_$i.txt 2>&1
    F=$(find $OUT/p -name "*counter_collection.csv" | head -1); cp "$F" $OUT/sq_${m}_$i.csv; rm -rf $OUT/p
  done
  python3 $R/tools/pmc_sq.py $OUT/sq_${m}_1.csv $OUT/sq_${m}_2.csv | grep -E "^tile_emit" > $OUT/emit_mode$m.txt
  cat $OUT/emit_mode$m.txt
done
