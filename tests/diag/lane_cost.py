"""Tail or divergence?  For every wave iteration of k_pair (64 pairs in processing order) compare the time the wave took on
the device with what its 64 pairs cost one by one on the CPU (the same kernel bodies, tests/hostemu.cpp): if the wave time
follows the LONGEST pair the lanes run in lockstep and the loss is the tail; if it follows the SUM the lanes are serialised
(control divergence).  Least-squares fit  wave_us = a * max(cost) + b * sum(cost) + c."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['CM_LANE_CLK'] = '1'
from circminer_amd import lib as cl, synth
import conftest
from oracle import oracle_py as op
N = int(os.environ.get('PAIRS', '131072'))
d = synth.generate(os.environ.get('PRESET', 'chr21'), n_pairs=N, seed=int(os.environ.get('SEED', '21')))
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs[:1], [t for t in d.chr_table if t[1] == 1], '/tmp/c.gtf', n_threads=16)
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
hp.L.cm_debug_lane_clk.argtypes = [C.c_void_p, C.c_void_p]
b = cl.ReadBatch(d.seq1, d.seq2); hp.upload(b)
hp.reset(); hp.map_round(0, True); hp.sync()
clk = np.zeros(b.n, np.uint64); assert hp.L.cm_debug_lane_clk(hp.h, clk.ctypes.data) == 0
wave_us = (clk & np.uint64(0xFFFFFFFF)).astype(np.float64) / 100.0
pair = (clk >> np.uint64(32)).astype(np.int64)
n_l = int(np.nonzero(wave_us)[0].max()) + 1
n_w = n_l // 64
wave_us, pair = wave_us[:n_w * 64].reshape(-1, 64), pair[:n_w * 64].reshape(-1, 64)
E = conftest.load_emu()
E.emu_set_cost_out.argtypes = [C.c_void_p]
cost = np.zeros(b.n, np.float64); E.emu_set_cost_out(cost.ctypes.data)
op.build()
st, act = op.default_state(P, b.n); cat = np.full(b.n, -1, np.int32)
assert E.emu_map_round(C.byref(P), C.byref(hi.views[0]), C.byref(hi.annots[0]), C.byref(b.c), 1, st.ctypes.data, act.ctypes.data, cat.ctypes.data) == 0
E.emu_set_cost_out(None)
cu = cost[pair] / 1000.0                                  # us on one CPU core, per lane
wt = wave_us[:, 0]
mx, sm = cu.max(1), cu.sum(1)
A = np.stack([mx, sm, np.ones_like(mx)], 1)
coef, *_ = np.linalg.lstsq(A, wt, rcond=None)
pred = A @ coef
print('wave iterations %d; device wave time mean %.0f us; CPU cost per pair mean %.1f us (max of 64: %.0f, sum of 64: %.0f)' % (n_w, wt.mean(), cu.mean(), mx.mean(), sm.mean()))
print('fit wave_us = %.2f * max + %.3f * sum + %.0f   (R^2 %.3f)' % (coef[0], coef[1], coef[2], 1 - ((wt - pred) ** 2).sum() / ((wt - wt.mean()) ** 2).sum()))
print('share of the fitted time: max term %.0f%%, sum term %.0f%%, constant %.0f%%' % tuple(100 * x / pred.mean() for x in (coef[0] * mx.mean(), coef[1] * sm.mean(), coef[2])))
print('corr(wave, max) %.3f  corr(wave, sum) %.3f' % (np.corrcoef(wt, mx)[0, 1], np.corrcoef(wt, sm)[0, 1]))
print('cost fill of the waves as ordered now: sum / (64 * max) = %.3f;  with pairs sorted by CPU cost: %.3f' % (sm.sum() / (64 * mx.sum()), (lambda s: s.sum() / (64 * s.reshape(-1, 64).max(1).sum()))(np.sort(cu.ravel()))))
order = np.argsort(mx)
print('waves binned by the CPU cost of their longest pair:')
for k in range(10):
    sel = order[k * n_w // 10:(k + 1) * n_w // 10]
    print('  decile %d: max %.2f us  sum %.1f us  -> device wave %.0f us (min %.0f)   categories %s' % (k, mx[sel].mean(), sm[sel].mean(), wt[sel].mean(), wt[sel].min(),
          np.bincount(np.clip(cat[pair[sel]].ravel(), 0, 15), minlength=12)[:12].tolist()))
print('position in the processing order vs wave time (first to last):')
for k in range(10):
    sel = np.arange(k * n_w // 10, (k + 1) * n_w // 10)
    print('  part %d: device wave %.0f us   max %.2f  sum %.1f' % (k, wt[sel].mean(), mx[sel].mean(), sm[sel].mean()))
