#!/bin/bash
# Round-4 profile evidence, in two gpurun calls (a call is limited to 20 minutes):
#   bash profiles/collect_r04.sh bench   -> kernel stats of the default bench command, FETCH_SIZE / WRITE_SIZE passes, traffic.json, the bench line
#   bash profiles/collect_r04.sh mix     -> SQ_* passes of one 2^20-pair round on the dense genome (tests/diag/pmc_mix.sh) -> pmc_mix.json + summary
# Outputs land in gpurun_out/profiles/ (merged back by gpurun); copy into profiles/ afterwards.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
if [ "$1" = "bench" ]; then
  bash profiles/collect.sh r04
elif [ "$1" = "mix" ]; then
  bash tests/diag/pmc_mix.sh gpurun_out/profiles/mix_r04 PRESET=hg38like PAIRS=1048576 REPS=3 > $OUT/r04_pmc_mix_summary.txt 2>&1
  cp $OUT/mix_r04/pmc_mix.json $OUT/r04_pmc_mix.json
  tail -5 $OUT/r04_pmc_mix_summary.txt
fi
