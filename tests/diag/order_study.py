"""How much of k_pair's time is control divergence?  Re-run the same reads in an order derived from the
RESULTS of a first run (an oracle of each pair's path) and compare k_pair's time."""
import os, sys, tempfile, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import lib as cl, synth
N = 1_000_000
d = synth.generate('chr21', n_pairs=N, seed=21)
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs, d.chr_table, '/tmp/c.gtf', n_threads=16)
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
def run(order, label):
    b = cl.ReadBatch(d.seq1[order], d.seq2[order]); hp.upload(b)
    for it in range(3):
        if it == 1: hp.prof(True); hp.prof_reset()
        hp.reset(); hp.map_round(0, True); hp.sync()
    ms, n, cnt = hp.prof_get(); hp.prof(False)
    print('%-34s k_pair %.2f ms  k_pair_heavy %.2f  k_chain %.2f  k_seed %.2f' % (label, ms[2] / 2, ms[4] / 2, ms[1] / 2, ms[0] / 2), flush=True)
    return hp.download()
ident = np.arange(N)
st, cat, act = run(ident, 'generated order')
rng = np.random.default_rng(3)
run(rng.permutation(N), 'random order')
ed = st['ed_r1'].astype(np.int64) + st['ed_r2']
keys = {
    'by type': st['type'].astype(np.int64),
    'by type, ed': st['type'].astype(np.int64) * 64 + np.minimum(ed, 63),
    'by type, ed, junc': (st['type'].astype(np.int64) * 64 + np.minimum(ed, 63)) * 8 + np.minimum(st['junc_num'], 7),
    'by type, ed_r1, ed_r2, junc, mlen': ((((st['type'].astype(np.int64) * 16 + np.minimum(st['ed_r1'], 15)) * 16 + np.minimum(st['ed_r2'], 15)) * 8 + np.minimum(st['junc_num'], 7)) * 512
                                          + st['mlen_r1'].astype(np.int64)) * 512 + st['mlen_r2'],
}
for k, v in keys.items():
    run(np.argsort(v, kind='stable'), k)
