#!/bin/bash
# k_hp_tasks' time per launch with and without the task order (kernel trace of one 2^20-pair round, tests/diag/pcs_run.py)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/tasks_time; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PRESET=hg38like PAIRS=1048576 REPS=3
for v in on off; do
  if [ $v = off ]; then export CM_HP_TASK_ORDER=0; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/$v -o st --output-format csv -- python3 $ROOT/tests/diag/pcs_run.py > $OUT/$v.log 2>&1
  echo "== order $v"; grep -E "k_hp_tasks|k_hp_dp|k_hp_plan|k_cls_place|k_pair\(" $OUT/$v/st_kernel_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://; s/(.*)"/"/' 
done
