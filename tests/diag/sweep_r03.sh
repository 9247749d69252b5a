#!/bin/bash
# bench.py under a few settings of one tuning knob: sweep_r03.sh VAR v1 v2 ...   (results: value, ms per step)
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],1), 'ms', {k:(round(v['ms_total']/max(v['launches'],1),2)) for k,v in d['kernels'].items() if k in ('k_chain','k_pair','k_pair_heavy','k_chain_heavy')})"
done
