#!/bin/bash
# L2 (TCC) view of the memory-side kernels (k_seed, k_chain): requests, hits / misses, fabric reads and writes by size.
# Separate passes of a few counters each; TCP_* groups are left out (that pass aborted the profiler on this pool).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_l2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o pmc --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -2 $OUT/p$i.log; exit 1; }
  echo "pass $i done"
done
cd $ROOT
python3 - <<'PY'
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_l2/p*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[1] if r["Kernel_Name"].startswith("(anonymous") else r["Kernel_Name"]
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_l2/summary.txt", "w") as out:
    for k in ("k_seed", "k_chain", "k_chain_heavy", "k_pair", "k_pair_heavy"):
        if k in agg:
            line = k + ": " + ", ".join(f"{c}={sum(v) / len(v):.3g}" for c, v in sorted(agg[k].items()))
            print(line); out.write(line + "\n")
PY
