set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ablate2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in engine base; do
  if [ $v = base ]; then export CM_LIB=$ROOT/tests/_hostemu/libcmhot_base.so; else export CM_LIB=$ROOT/circminer_amd/csrc/libcmhot_engine.so; fi
  PRESET=hg38like PAIRS=262144 REPS=2 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES -d $OUT/$v -o pmc --output-format csv -- python3 $ROOT/tests/diag/pcs_run.py > $OUT/$v.log 2>&1 || { echo "$v failed"; tail -3 $OUT/$v.log; }
  python3 - $OUT/$v $v <<'PY'
import csv, glob, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + '/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r'\bk_\w+', r['Kernel_Name']); k = m.group(0) if m else '?'
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in ('k_pair_heavy', 'k_pair'):
    a = acc[k]; L = max(len(n[k]), 1)
    if a: print('%-8s %-14s VALU wave-instr/launch %8.1f M   lanes/instr %5.1f   wave cycles %8.1f M  valu busy %.3f' % (sys.argv[2], k, a['SQ_INSTS_VALU'] / L / 1e6, a['SQ_THREAD_CYCLES_VALU'] / max(a['SQ_ACTIVE_INST_VALU'], 1), a['SQ_WAVE_CYCLES'] / L / 1e6, a['SQ_ACTIVE_INST_VALU']/max(a['SQ_WAVE_CYCLES'],1)))
PY
done
