#!/bin/bash
# A/B of a build flag: ab_lib.sh TAG "-DFLAG ..." WORKLOAD   (bench.py with the product library, then with the variant)
TAG=$1; FLAGS=$2; WL=${3:-hg38like}
SO=$(python -c "from circminer_amd import _build; print(_build.build(tag='$TAG', flags='$FLAGS'.split()))")
for lib in "" "$SO"; do
  env CM_LIB=$lib python bench.py --workload $WL --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$WL', '${lib:-product}', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],1), 'ms', {k:(round(v['ms_total']/max(v['launches'],1),2)) for k,v in d['kernels'].items() if k in ('k_chain','k_pair','k_pair_heavy','k_chain_heavy')})"
done
