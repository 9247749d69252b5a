#!/bin/bash
# sweep of the pair-stage class layout (diagnostic; run on the GPU box)
for cfg in "8 0" "8 1" "8 2" "12 1" "12 2" "16 2"; do
  set -- $cfg
  CM_HEAVY_COST=$1 CM_CLS_MODE=$2 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/sw.json 2>gpurun_out/sw.err || exit 1
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sw.json"))
print(sys.argv[1], round(d["value"]/1e6,2), round(d["ms_per_step"],2), {k:round(v["ms_total"]/5,2) for k,v in d["kernels"].items() if "pair" in k or "classify" in k})
PY
done
