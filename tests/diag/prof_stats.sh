set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_r02a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o st --output-format csv -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
ls $OUT/stats | head
cat $OUT/stats/st_kernel_stats.csv | cut -c1-200
