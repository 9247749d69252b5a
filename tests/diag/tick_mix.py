"""Per-section wave time of k_pair (diag build) on the real read mix."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['CM_LIB'] = os.path.join(ROOT, 'tests/_hostemu/libcmhot_diag.so'); os.environ['CM_LANE_CLK'] = '1'
from circminer_amd import lib as cl, synth
N = 262144
d = synth.generate('chr21', n_pairs=N, seed=21)
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs, d.chr_table, '/tmp/c.gtf')
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
hp.L.cm_debug_lane_clk.argtypes = [C.c_void_p, C.c_void_p]
b = cl.ReadBatch(d.seq1, d.seq2); hp.upload(b); ch, nc, hh = hp.chains(0); nc = nc.reshape(-1, 4)
hp.reset(); hp.map_round(0, True); hp.sync()
clk = np.zeros(b.n * 16, np.uint64); assert hp.L.cm_debug_lane_clk(hp.h, clk.ctypes.data) == 0
st = hp.download()[0]; a = clk.reshape(-1, 16) / 100.0
names = {0: 'before process_mates (prologue / previous leftovers)', 14: 'leftover extensions', 1: 'pass1(pairing)', 2: 'pre-ext', 3: 'is_left', 4: 'middle_ed+concord', 5: 'chain_left l', 6: 'chain_left r', 7: 'chain_right r',
         8: 'chain_right l', 9: 'overlaps', 10: 'fold', 11: 'ext: trans loop', 12: 'ext: intron-ret DP', 13: 'tail(leftovers..)', 15: 'TOTAL'}
cost = nc[:, 0] * nc[:, 3] + nc[:, 2] * nc[:, 1] + nc.sum(1)
light = cost <= 8
print('light fraction', light.mean())
# per-wave: take the max over lanes of TOTAL (lanes of heavy pairs have 0)
tot = a[:, 15].reshape(-1, 64).max(1)
print('wave total us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f' % (tot.mean(), *np.percentile(tot, [50, 90, 99]), tot.max()))
# per-section: per wave take the lane with the largest total as representative
rep = a.reshape(-1, 64, 16)[np.arange(len(tot)), a[:, 15].reshape(-1, 64).argmax(1)]
for k in sorted(names): print('  %-20s mean %8.1f us  share %.2f' % (names[k], rep[:, k].mean(), rep[:, k].mean() / rep[:, 15].mean()))
# which pair types sit in the slow waves
slow = np.nonzero(tot > np.percentile(tot, 90))[0]
print('types in slowest 10% waves:', np.bincount(st['type'].reshape(-1, 64)[slow].ravel(), minlength=14))
print('types overall            :', np.bincount(st['type'], minlength=14))
# per-lane own work is not observable (lanes wait for each other); correlate wave time with presence of types
for t in range(14):
    has = (st['type'].reshape(-1, 64) == t).any(1)
    if has.sum() > 20 and (~has).sum() > 20: print('type %2d present: mean wave %.0f us (n=%d) | absent: %.0f us' % (t, tot[has].mean(), has.sum(), tot[~has].mean()))
