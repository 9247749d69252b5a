#!/bin/bash
# Instruction mix of the kernels (separate --pmc passes, kernel-trace only).  usage: pmc_mix.sh <out-dir> [env for pcs_run.py ...]
set -e
ROOT=$(pwd); OUT=$ROOT/$1; shift
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
i=0
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_FLAT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL")
if [ -n "$PASSES" ]; then SETS=("$PASSES"); fi     # PASSES="C1 C2 C3 C4": one pass with these counters
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc --output-format csv -- python3 $ROOT/tests/diag/pcs_run.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
  echo "pass $i done"
done
cd $ROOT
python3 tests/diag/pmc_mix.py $OUT
