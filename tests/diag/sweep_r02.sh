#!/bin/bash
# knob sweep on the hg38-like bench (5 steps each): chain light/heavy split, pair heavy cost
cd $(dirname $0)/../..
for kv in "A=1" "CM_CHAIN_LIGHT_W=1024" "CM_CHAIN_LIGHT_W=4096 CM_CHAIN_LIGHT_CELLS=128" "CM_HEAVY_COST=16" "CM_HEAVY_COST=40"; do
  echo "== $kv"
  env $kv python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1])
k = j['kernels']
print('value %.2f M pairs/s  ms/step %.2f' % (j['value'] / 1e6, j['ms_per_step']), {n: round(v['ms_total'] / j['steps'], 2) for n, v in k.items()})
"
done
