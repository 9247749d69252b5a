"""Per-section wave time of k_pair (diag build), 64 different pairs per wave vs 64 copies of one pair."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['CM_LIB'] = os.path.join(ROOT, 'tests/_hostemu/libcmhot_diag.so'); os.environ['CM_LANE_CLK'] = '1'
from circminer_amd import lib as cl, synth
d = synth.generate('chr21', n_pairs=100000, seed=21)
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs, d.chr_table, '/tmp/c.gtf')
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
hp.L.cm_debug_lane_clk.argtypes = [C.c_void_p, C.c_void_p]
def run(idx):
    b = cl.ReadBatch(d.seq1[idx], d.seq2[idx]); hp.upload(b); hp.map_round(0, True); hp.sync()
    clk = np.zeros(b.n * 16, np.uint64); assert hp.L.cm_debug_lane_clk(hp.h, clk.ctypes.data) == 0
    return hp.download()[0], clk.reshape(-1, 16) / 100.0
b = cl.ReadBatch(d.seq1, d.seq2); hp.upload(b); ch, nc, hh = hp.chains(0); nc = nc.reshape(-1, 4)
st, _ = run(np.arange(100000))
names = {1: 'pass1(pairing)', 2: 'pre-ext', 3: 'is_left', 4: 'middle_ed+concord', 5: 'chain_left l', 6: 'chain_left r', 7: 'chain_right r',
         8: 'chain_right l', 9: 'overlaps', 10: 'fold', 11: 'ext: trans loop', 12: 'ext: intron-ret DP', 13: 'tail', 15: 'TOTAL'}
for label, sel in [('simple CONCRD', (nc[:, 0] == 1) & (nc[:, 1] == 0) & (nc[:, 2] == 0) & (nc[:, 3] == 1) & (st['type'] == 0) & (st['ed_r1'] + st['ed_r2'] == 0) & (st['junc_num'] == 0)),
                   ('simple CONGNM', (nc[:, 0] + nc[:, 3] == 2) & (nc[:, 1] + nc[:, 2] == 0) & (st['type'] == 7) & (st['ed_r1'] + st['ed_r2'] == 0))]:
    ids = np.nonzero(sel)[0][:8192]
    _, a = run(ids); _, bb = run(np.repeat(ids[:128], 64))
    A = a[::64].mean(0); B = bb[::64].mean(0)
    print('==', label, len(ids))
    for k in sorted(names): print('  %-20s different %8.1f us   copies %8.1f us   x%.1f' % (names[k], A[k], B[k], A[k] / max(B[k], 1e-9)))
