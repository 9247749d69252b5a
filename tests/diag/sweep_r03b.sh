#!/bin/bash
run() { env "$@" python bench.py --workload ${WL:-hg38like} --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-hg38like}', '$*', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],1), 'ms', {k:(round(v['ms_total']/max(v['launches'],1),2)) for k,v in d['kernels'].items() if k in ('k_chain','k_pair','k_pair_heavy','k_chain_heavy')})"; }
run X=1
run CM_CHAIN_LIGHT_W=1024 CM_CHAIN_LIGHT_CELLS=192
run CM_CHAIN_LIGHT_W=4096 CM_CHAIN_LIGHT_CELLS=256
run CM_CHAIN_LIGHT_W=64 CM_CHAIN_LIGHT_CELLS=48
WL=chr21 run X=1
WL=hg38like_sparse run X=1
