#!/bin/bash
# knob sweep on the hg38-like bench (5 steps each); arguments = "VAR=val[ VAR=val]" settings to try
cd $(dirname $0)/../..
for kv in "A=1" "$@"; do
  echo "== $kv"
  env $kv python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1])
k = j['kernels']
print('value %.2f M pairs/s  ms/step %.2f' % (j['value'] / 1e6, j['ms_per_step']), {n: round(v['ms_total'] / j['steps'], 2) for n, v in k.items()})
"
done
