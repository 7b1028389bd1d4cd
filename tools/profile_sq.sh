#!/bin/bash
# SQ counter passes of the bench command (separate rocprofv3 --pmc runs, --kernel-trace only):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/profile_sq.sh r02 sq'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}_${2:-sq}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*\|SQ_LDS[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_WAVES" $OUT/counters_list.txt | sort -u > $OUT/sq_names.txt
wc -l $OUT/sq_names.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" ; do
  i=$((i+1))
  echo "[sq] pass $i: $set"
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stage-profile > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; }
  F=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && cp "$F" $OUT/pmc_sq_$i.csv
  rm -rf $OUT/p$i
done
python3 $R/tools/pmc_sq.py $OUT/pmc_sq_*.csv > $OUT/pmc_sq.txt 2>&1
head -40 $OUT/pmc_sq.txt
