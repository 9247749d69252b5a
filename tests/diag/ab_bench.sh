#!/bin/bash
# A/B of two builds of the library on the bench: ab_bench.sh <variant .so> [bench args]
cd $(dirname $0)/../..
V=$1; shift
fmt='
import sys, json
j = json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print("%s value %.2f M pairs/s  ms/step %.2f" % (sys.argv[1], j["value"] / 1e6, j["ms_per_step"]), {n: round(v["ms_total"] / j["steps"], 2) for n, v in j["kernels"].items()})
'
for rep in 1 2; do
python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" default
CM_LIB=$V python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" variant
done
