"""Driver for the PMC passes of pmc_mix.sh: a few mapping rounds of one batch on chr21.
BATCH=mix|conc|same|rep64|rest  PRESET  PAIRS  REPS  CM_LIB (a variant build)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import lib as cl, synth
N = int(os.environ.get('PAIRS', '1048576'))
d = synth.generate(os.environ.get('PRESET', 'chr21'), n_pairs=N, seed=21)
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs[:1], [t for t in d.chr_table if t[1] == 1], '/tmp/c.gtf', n_threads=16)
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
b = cl.ReadBatch(d.seq1, d.seq2); hp.upload(b)
hp.reset(); hp.map_round(0, True); hp.sync()
B = os.environ.get('BATCH', 'mix')
if B != 'mix':
    cat = hp.download()[1]
    sel = np.nonzero(cat == 0)[0]
    if B == 'conc': sel = sel[np.arange(N) % len(sel)]             # concordant pairs only, all different
    elif B == 'same': sel = np.repeat(sel[:1], N)                   # one pair N times
    elif B == 'rep64': sel = np.repeat(sel[:N // 64], 64)           # each wave = 64 copies of one pair
    elif B == 'rest': sel = np.nonzero(cat != 0)[0]; sel = sel[np.arange(N) % len(sel)]
    b = cl.ReadBatch(d.seq1[sel], d.seq2[sel]); hp.upload(b)
for _ in range(int(os.environ.get('REPS', '5'))):
    hp.reset(); hp.map_round(0, True); hp.sync()
print('done')
