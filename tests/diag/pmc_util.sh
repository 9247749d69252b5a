#!/bin/bash
# lane utilisation of the VALU instructions per kernel (diagnostic; run on the GPU box)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES -d /root/repo/gpurun_out/pmc_util -o pmc --output-format csv -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_util.log 2>&1 || { echo failed; tail -3 /root/repo/gpurun_out/pmc_util.log; }
