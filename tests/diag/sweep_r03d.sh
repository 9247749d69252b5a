#!/bin/bash
# pairs per wave of k_pair_heavy (library variants built with -DCM_HEAVY_G=n) x light / heavy threshold: sweep_r03d.sh "CM_LIB=... CM_HEAVY_COST=6" ...
for kv in "$@"; do
  env $kv python bench.py --workload ${WL:-hg38like} --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-hg38like}', '$kv', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],1), 'ms', {k:(round(v['ms_total']/max(v['launches'],1),2)) for k,v in d['kernels'].items() if k in ('k_chain','k_pair','k_pair_heavy','k_chain_heavy')})"
done
