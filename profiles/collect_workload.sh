#!/bin/bash
# collect.sh for another workload of bench.py (e.g. hg38like_sparse, the round-2 genome): kernel stats, FETCH_SIZE / WRITE_SIZE
# passes and the bench line, written to gpurun_out/profiles/<tag>_*; copy what should be judged into profiles/.
#   usage (through gpurun, from the repo root): bash profiles/collect_workload.sh r03_sparse hg38like_sparse
set -e
R=${1:-r03_sparse}; WL=${2:-hg38like_sparse}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats_$R -o st --output-format csv -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline > $OUT/stats_$R.log 2>&1
cp $OUT/stats_$R/st_kernel_stats.csv $OUT/${R}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$R -o pmc --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch_$R.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$R -o pmc --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write_$R.log 2>&1
cd $ROOT
python3 profiles/make_traffic.py $OUT/fetch_$R/pmc_counter_collection.csv $OUT/write_$R/pmc_counter_collection.csv $OUT/${R}_traffic.json $OUT/${R}_pmc_fetch_size.csv $OUT/${R}_pmc_write_size.csv
python3 bench.py --workload $WL > $OUT/${R}_bench_1gpu.json 2> $OUT/bench_$R.err
python3 -c "
import json; d=json.load(open('$OUT/${R}_bench_1gpu.json')); print('$WL', d['value'], d['ms_per_step'], d['roofline']['frac'])
t=json.load(open('$OUT/${R}_traffic.json')); print(t['bytes_per_launch'])"
