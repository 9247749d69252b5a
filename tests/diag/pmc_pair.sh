#!/bin/bash
# PMC passes for the pair stage (diagnostic; run on the GPU box).  One counter group per pass.
# (A pass with TCP_*_sum derived counters aborted rocprofv3 on this pool and hung the call: do not add one.)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVE_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d /root/repo/gpurun_out/pmc_pair$i -o pmc --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_pair$i.log 2>&1 || { echo "pass $i failed"; tail -3 /root/repo/gpurun_out/pmc_pair$i.log; }
done
