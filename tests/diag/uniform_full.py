"""Pair-stage time of a full 1 M-pair batch when (a) the batch is the synthetic mix, (b) concordant pairs only, all
different, (c) one concordant pair 1 M times (perfect control AND memory convergence), (d) 16 K different pairs each 64
times in a row (control convergence, memory divergence across waves only).  cm_prof stage timers of the product build."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import lib as cl, synth
N = int(os.environ.get('PAIRS', '1048576'))
d = synth.generate(os.environ.get('PRESET', 'chr21'), n_pairs=N, seed=21)
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs[:1], [t for t in d.chr_table if t[1] == 1], '/tmp/c.gtf', n_threads=16)
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])

def run(s1, s2, label):
    b = cl.ReadBatch(s1, s2); hp.upload(b)
    for rep in range(3):
        hp.reset(); hp.prof(True); hp.prof_reset(); hp.map_round(0, True); hp.sync()
        ms, n, cnt = hp.prof_get()
    st, cat, act = hp.download()
    print('%-52s seed %.2f chain %.2f pair %.2f (k_pair %.2f, k_pair_heavy %.2f, chain_heavy %.2f) ms   cats %s' % (label, ms[0], ms[1], ms[2], ms[2], ms[4], ms[6], np.bincount(np.clip(cat, 0, 11), minlength=12)[:8].tolist()))
    return cat

cat = run(d.seq1, d.seq2, 'the synthetic batch as it is')
exact = np.nonzero(cat == 0)[0]
sel = exact[np.arange(N) % len(exact)]
run(d.seq1[sel], d.seq2[sel], 'concordant pairs only, all different')
i = exact[0]
run(np.repeat(d.seq1[i:i + 1], N, 0), np.repeat(d.seq2[i:i + 1], N, 0), 'one concordant pair x N')
base = exact[:N // 64]
sel = np.repeat(base, 64)
run(d.seq1[sel], d.seq2[sel], 'N/64 concordant pairs x 64 copies in a row')
rest = np.nonzero(cat != 0)[0]
sel = rest[np.arange(N) % len(rest)]
run(d.seq1[sel], d.seq2[sel], 'everything but the concordant pairs')
