"""Intrinsic k_pair time of single pairs (each replicated over a whole wave) vs features known before k_pair."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['CM_LIB'] = os.path.join(ROOT, 'tests/_hostemu/libcmhot_diag.so'); os.environ['CM_LANE_CLK'] = '1'
from circminer_amd import lib as cl, synth
N = 200000
d = synth.generate('chr21', n_pairs=N, seed=21)
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs, d.chr_table, '/tmp/c.gtf')
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
hp.L.cm_debug_lane_clk.argtypes = [C.c_void_p, C.c_void_p]
b = cl.ReadBatch(d.seq1, d.seq2); hp.upload(b); ch, nc, hh = hp.chains(0); nc = nc.reshape(-1, 4)
ch = ch.reshape(-1, 4, 30)
cost = nc[:, 0] * nc[:, 3] + nc[:, 2] * nc[:, 1] + nc.sum(1)
rng = np.random.default_rng(1)
ids = rng.choice(np.nonzero(cost <= 8)[0], 3000, replace=False)
rep = np.repeat(ids, 64)
bb = cl.ReadBatch(d.seq1[rep], d.seq2[rep])
hp.upload(bb); hp.map_round(0, True); hp.sync()
clk = np.zeros(bb.n * 16, np.uint64); assert hp.L.cm_debug_lane_clk(hp.h, clk.ctypes.data) == 0
st = hp.download()[0][::64]
t = (clk.reshape(-1, 16)[:, 15] / 100.0)[::64]
k = 20
def resid(c, n, L=150):
    r = np.zeros(len(c))
    for x in range(len(c)):
        if n[x] > 0:
            cl_ = c[x, 0]['chain_len']; r[x] = c[x, 0]['qpos'][0] + (L - (c[x, 0]['qpos'][cl_ - 1] + k))
    return r
rs = np.stack([resid(ch[ids, s], nc[ids, s]) for s in range(4)], 1)
two = ((nc[ids, 0] > 0) & (nc[ids, 3] > 0)).astype(int) + ((nc[ids, 2] > 0) & (nc[ids, 1] > 0)).astype(int)
print('intrinsic us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f' % (t.mean(), *np.percentile(t, [50, 90, 99]), t.max()))
print('by final type:')
for ty in np.unique(st['type']):
    m = st['type'] == ty
    print('  type %2d n=%4d mean %7.0f p90 %7.0f' % (ty, m.sum(), t[m].mean(), np.percentile(t[m], 90)))
print('by cost:')
for cst in range(0, 9):
    m = cost[ids] == cst
    if m.sum(): print('  cost %d n=%4d mean %7.0f p90 %7.0f' % (cst, m.sum(), t[m].mean(), np.percentile(t[m], 90)))
print('by resid sum bucket:')
R = rs.sum(1)
for lo, hi_ in [(0, 1), (1, 25), (25, 50), (50, 100), (100, 150), (150, 200), (200, 300), (300, 400), (400, 1000)]:
    m = (R >= lo) & (R < hi_)
    if m.sum(): print('  resid [%3d,%3d) n=%4d mean %7.0f p50 %7.0f p90 %7.0f max %7.0f' % (lo, hi_, m.sum(), t[m].mean(), np.percentile(t[m], 50), np.percentile(t[m], 90), t[m].max()))
print('by #orientations with both mates chained:', [(x, int((two == x).sum()), round(float(t[two == x].mean()), 0)) for x in range(3) if (two == x).sum()])
ed = st['ed_r1'].astype(int) + st['ed_r2']
print('by edits (CONCRD/CONGNM only):')
for e in range(0, 6):
    m = (ed == e) & np.isin(st['type'], [0, 7])
    if m.sum(): print('  ed %d n=%4d mean %7.0f' % (e, m.sum(), t[m].mean()))
print('junc_num:', [(j, int((st['junc_num'] == j).sum()), round(float(t[st['junc_num'] == j].mean()), 0)) for j in np.unique(st['junc_num'])])
# simple linear fit on features
X = np.stack([np.ones(len(ids)), cost[ids], R, two, nc[ids].sum(1)], 1)
w, *_ = np.linalg.lstsq(X, t, rcond=None)
print('lstsq [1, cost, resid, two, nchains] ->', np.round(w, 1), ' R2 = %.2f' % (1 - ((X @ w - t) ** 2).sum() / ((t - t.mean()) ** 2).sum()))
np.save(os.path.join(ROOT, 'gpurun_out', 'pair_cost.npy'), np.column_stack([ids, t, cost[ids], R, two, st['type'], ed, st['junc_num']]))
