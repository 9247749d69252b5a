"""Sums the counters of pmc_mix.sh per kernel (per launch)."""
import csv, glob, os, re, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in glob.glob(os.path.join(out, 'p*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        m = re.search(r'\bk_\w+', r['Kernel_Name'])
        k = m.group(0) if m else r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        launches[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_INSTS_VALU", 0))[:14]:
    print(k)
    for c in sorted(acc[k]):
        n = max(len(launches[(k, c)]), 1)
        print('   %-26s %16.0f per launch' % (c, acc[k][c] / n))
    a = acc[k]
    if a.get('SQ_ACTIVE_INST_VALU'):
        print('   lanes per VALU instruction   %.1f' % (a['SQ_THREAD_CYCLES_VALU'] / a['SQ_ACTIVE_INST_VALU']))
    if a.get('SQ_BUSY_CYCLES') and a.get('SQ_ACTIVE_INST_VALU'):
        # SQ_ACTIVE_INST_VALU: cycles (x4, per SIMD quad-cycle) a VALU instruction was executing; SQ_WAVE_CYCLES: wave-resident cycles
        print('   VALU busy / wave cycles      %.3f' % (a['SQ_ACTIVE_INST_VALU'] / max(a.get('SQ_WAVE_CYCLES', 0), 1)))
    if a.get('SQ_LDS_IDX_ACTIVE'):
        print('   LDS bank-conflict cycles / LDS active cycles   %.3f' % (a.get('SQ_LDS_BANK_CONFLICT', 0) / a['SQ_LDS_IDX_ACTIVE']))

# machine-readable summary for bench.py's `stages` object (copy to profiles/pmc_mix.json)
import json
js = {"workload": os.environ.get("PRESET", "chr21"), "pairs_per_launch": int(os.environ.get("PAIRS", "0") or 0),
      "source": "rocprofv3 --pmc passes of tests/diag/pmc_mix.sh (one mapping round of one tile against packed contig 0; counters summed per kernel, per launch)",
      "kernels": {}}
for k, a in acc.items():
    if not k.startswith("k_"):
        continue
    e = {}
    n = lambda c: max(len(launches[(k, c)]), 1)
    if a.get("SQ_ACTIVE_INST_VALU"):
        e["lanes_per_valu_inst"] = round(a["SQ_THREAD_CYCLES_VALU"] / a["SQ_ACTIVE_INST_VALU"], 2)
        e["valu_wave_insts_per_launch"] = round(a.get("SQ_INSTS_VALU", 0) / n("SQ_INSTS_VALU"))
    if a.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_rate"] = round(a.get("SQ_LDS_BANK_CONFLICT", 0) / a["SQ_LDS_IDX_ACTIVE"], 4)
        e["lds_insts_per_launch"] = round(a.get("SQ_INSTS_LDS", 0) / n("SQ_INSTS_LDS"))
    if a.get("SQ_WAVE_CYCLES") and a.get("SQ_ACTIVE_INST_VALU"):
        e["valu_busy_of_wave_cycles"] = round(a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"], 3)
    if e:
        js["kernels"][k] = e
json.dump(js, open(os.path.join(out, "pmc_mix.json"), "w"), indent=1)
