#!/bin/bash
# Collects the round's profile evidence on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> profiles/rNN_kernel_stats.csv
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)               -> profiles/rNN_pmc_*.csv, profiles/traffic.json
#   3. the bench line itself (with cpu_baseline)                        -> profiles/rNN_bench_1gpu.json
# Outputs are written under gpurun_out/profiles/ (merged back by gpurun); copy them into profiles/ afterwards.
set -e
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o st --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
cp $OUT/stats/st_kernel_stats.csv $OUT/${R}_kernel_stats.csv
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
echo "write done"
cd $ROOT
python3 profiles/make_traffic.py $OUT/fetch/pmc_counter_collection.csv $OUT/write/pmc_counter_collection.csv $OUT/traffic.json $OUT/${R}_pmc_fetch_size.csv $OUT/${R}_pmc_write_size.csv
cp $OUT/traffic.json profiles/traffic.json
python3 bench.py > $OUT/${R}_bench_1gpu.json 2> $OUT/bench.err
cat $OUT/${R}_bench_1gpu.json
