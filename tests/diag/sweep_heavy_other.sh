for v in 6 24 48; do
  env CM_HEAVY_COST=$v python bench.py --workload chr21 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chr21 CM_HEAVY_COST=$v', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],1), 'ms')"
  env CM_HEAVY_COST=$v python bench.py --workload hg38like_sparse --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sparse CM_HEAVY_COST=$v', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],1), 'ms')"
done
