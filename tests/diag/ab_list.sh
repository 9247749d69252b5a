#!/bin/bash
# bench a list of "ENV=... ENV=..." settings (one per argument; CM_LIB=<path> selects a variant build); 6 steps each
cd $(dirname $0)/../..
for kv in "$@"; do
  echo "== $kv"
  env $kv python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1])
k = j['kernels']
print('value %.2f M pairs/s  ms/step %.2f' % (j['value'] / 1e6, j['ms_per_step']), {n: round(v['ms_total'] / j['steps'], 2) for n, v in k.items()})
"
done
