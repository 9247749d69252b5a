"""Distribution of heavy pairs by candidate chain pairs (diagnostic, GPU box)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import lib as cl, synth
N = 1_000_000
d = synth.generate('chr21', n_pairs=N, seed=21)
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs, d.chr_table, '/tmp/c.gtf', n_threads=16)
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
b = cl.ReadBatch(d.seq1, d.seq2); hp.upload(b)
ch, nc, hh = hp.chains(0); nc = nc.reshape(-1, 4).astype(np.int64)
a, bb, c, dd = nc[:, 0], nc[:, 1], nc[:, 2], nc[:, 3]
cost = a * dd + c * bb + nc.sum(1)
T = np.maximum(a * dd, c * bb)
heavy = cost > 8
print('heavy pairs', int(heavy.sum()), 'of', N)
for lo, hi_ in [(0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 256), (256, 512), (512, 901)]:
    m = heavy & (T >= lo) & (T < hi_)
    print('  T in [%3d,%3d): %6d pairs, sum chains %7d, sum T %8d' % (lo, hi_, m.sum(), nc[m].sum(), T[m].sum()))
