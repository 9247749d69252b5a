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
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_INSTS_VALU', 0))[:8]:
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
