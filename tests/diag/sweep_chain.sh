#!/bin/bash
# sweep of the light/heavy threshold of the chaining stage (diagnostic; run on the GPU box)
for cfg in "256 96" "1024 128" "4096 256" "16384 512"; do
  set -- $cfg
  CM_CHAIN_LIGHT_W=$1 CM_CHAIN_LIGHT_CELLS=$2 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/sw.json 2>gpurun_out/sw.err || exit 1
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sw.json"))
print(sys.argv[1], round(d["value"]/1e6,2), round(d["ms_per_step"],2), {k:round(v["ms_total"]/5,2) for k,v in d["kernels"].items() if "chain" in k or "classify" in k})
PY
done
