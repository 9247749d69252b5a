"""Offline study (CPU only): how well would candidate sort keys fill the light kernel's waves?  Per-pair cost = CPU time of the
pair stage in the host emulation of the kernel bodies; fill(order) = sum(cost) / (64 x sum over waves of the wave's largest cost)."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from circminer_amd import lib as cl, synth
import conftest
from oracle import oracle_py as op
N = int(os.environ.get('PAIRS', '65536'))
d = synth.generate(os.environ.get('PRESET', 'chr21'), n_pairs=N, seed=21)
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs[:1], [t for t in d.chr_table if t[1] == 1], '/tmp/c.gtf', n_threads=8)
P = cl.default_params(); k = P.kmer
b = cl.ReadBatch(d.seq1, d.seq2)
E = conftest.load_emu(); op.build()
iv, av = hi.views[0], hi.annots[0]
ch, nc, hh = op.chains(P, iv, av, b)
ch = ch.reshape(N, 4, cl.CM_BESTCHAINLIM); nc = nc.reshape(N, 4)
E.emu_set_cost_out.argtypes = [C.c_void_p]; E.emu_set_dp_out.argtypes = [C.c_void_p]
cost = np.zeros(N); dps = np.zeros(N, np.uint32)
best = np.full(N, 1e18)
for rep in range(3):
    E.emu_set_cost_out(cost.ctypes.data); E.emu_set_dp_out(dps.ctypes.data)
    st, act = op.default_state(P, N); cat = np.full(N, -1, np.int32)
    assert E.emu_map_round(C.byref(P), C.byref(iv), C.byref(av), C.byref(b.c), 1, st.ctypes.data, act.ctypes.data, cat.ctypes.data) == 0
    best = np.minimum(best, cost)
E.emu_set_cost_out(None); E.emu_set_dp_out(None)
cost = best / 1000.0
a, bb, cc, dd = nc[:, 0], nc[:, 1], nc[:, 2], nc[:, 3]
pc = a * dd + cc * bb + a + bb + cc + dd
light = ~(((a + bb) > 0) & ((cc + dd) > 0) & (pc > 6))
print('pairs %d, light %d; cost mean %.2f us (light %.2f, heavy %.2f)' % (N, light.sum(), cost.mean(), cost[light].mean(), cost[~light].mean() if (~light).any() else 0))

# features of the best chain of each problem
g = d.contigs[0]
L = d.seq1.shape[1]
first_q = ch['qpos'][:, :, 0, 0].astype(np.int64)
clen = ch['chain_len'][:, :, 0].astype(np.int64)
lastq = np.take_along_axis(ch['qpos'][:, :, 0, :].astype(np.int64), np.maximum(clen - 1, 0)[:, :, None], 2)[:, :, 0]
left = np.where(nc > 0, first_q, 0); right = np.where(nc > 0, L - (lastq + k), 0)
gaps = np.where(nc > 0, (lastq - first_q) // k + 1 - clen, 0)
main0 = (a > 0) & (dd > 0)
fx, bx = np.where(main0, 0, 2), np.where(main0, 3, 1)
idx = np.arange(N)
res4 = np.stack([left[idx, fx], right[idx, fx], left[idx, bx], right[idx, bx]], 1)     # the four extensions of the main orientation
mx = res4.max(1); tot = left.sum(1) + right.sum(1)
gap2 = gaps[idx, fx] + gaps[idx, bx]
ndp = dps.astype(np.int64)

def fill(order):
    c = cost[order]
    pad = (-len(c)) % 64
    w = np.concatenate([c, np.zeros(pad)]).reshape(-1, 64)
    return c.sum() / (64 * w.max(1).sum())

li = np.nonzero(light)[0]
rng = np.random.default_rng(1)
print('fill, light pairs:  random order %.3f   sorted by true cost %.3f' % (fill(rng.permutation(li)), fill(li[np.argsort(-cost[li], kind="stable")])))
lv = np.digitize(mx, [5, 10, 15, 20, 25, 30, 40, 50, 60, 70, 80, 90, 100, 115, 130])
pat = (res4 > 0) @ np.array([1, 2, 4, 8])
bucket = np.digitize(tot, [25, 50, 100, 150, 200, 300])
cur = np.lexsort((lv[li], pat[li], bucket[li]))          # class (bucket; genic flag not reproduced here) > pattern > longest residual
print('   ~current keys (total-residual bucket > extension pattern > longest residual): %.3f' % fill(li[cur][::-1]))
for name, key in (('longest residual', mx), ('total residual', tot), ('total residual + 40 x gaps', tot + 40 * gap2), ('number of real DPs (oracle knowledge)', ndp),
                  ('DPs x 64 + total residual', ndp * 64 + np.minimum(tot, 63)), ('chain-pair cost x 256 + total residual', pc * 256 + np.minimum(tot, 255))):
    print('   sorted by %-45s %.3f   corr with cost %.2f' % (name + ':', fill(li[np.argsort(-key[li], kind="stable")]), np.corrcoef(key[li], cost[li])[0, 1]))
# regression on available features
X = np.stack([np.ones(len(li)), tot[li], mx[li], gap2[li], pc[li], (res4[li] > 0).sum(1), ndp[li]], 1)
for cols, nm in (([0, 1, 2, 3, 4, 5], 'linear model on residuals, gaps, chain-pair cost, #extensions'), ([0, 1, 2, 3, 4, 5, 6], '... plus the number of real DPs')):
    w, *_ = np.linalg.lstsq(X[:, cols], cost[li], rcond=None)
    pred = X[:, cols] @ w
    print('   %-70s %.3f   corr %.2f' % (nm + ':', fill(li[np.argsort(-pred, kind="stable")]), np.corrcoef(pred, cost[li])[0, 1]))

# --- what k_pair_cls could compute itself: is each of the four residuals of the main orientation's best chains equal to the genome next to the chain?
comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
first_r = ch['rpos'][:, :, 0, 0].astype(np.int64)
last_r = np.take_along_axis(ch['rpos'][:, :, 0, :].astype(np.int64), np.maximum(clen - 1, 0)[:, :, None], 2)[:, :, 0]
inex = np.zeros((N, 2, 2), bool)        # [pair, which of the two main problems, left/right]
for side, xsel in ((0, fx), (1, bx)):
    for p in li:
        x = xsel[p]
        if nc[p, x] <= 0: continue
        mate, rc = x >> 1, x & 1
        rd = (d.seq2 if mate else d.seq1)[p]
        if rc: rd = comp[rd[::-1]]
        q0, r0 = first_q[p, x], first_r[p, x]          # rpos is 1-based: genome[r0 - 1] aligns with rd[q0]
        qe, re = lastq[p, x] + k, last_r[p, x] + k      # first base after the chain
        if q0 > 0:
            lo = r0 - 1 - q0
            inex[p, side, 0] = lo < 0 or not np.array_equal(g[lo:lo + q0], rd[:q0])
        if qe < L:
            lo = re - 1
            inex[p, side, 1] = lo + (L - qe) > len(g) or not np.array_equal(g[lo:lo + L - qe], rd[qe:])
nin = inex.reshape(N, 4).sum(1)
print('pairs by number of inexact residuals (genomic compare):', np.bincount(nin[li]).tolist(), ' corr(nin, real DPs) %.2f' % np.corrcoef(nin[li], ndp[li])[0, 1])
inlen = (inex.reshape(N, 4) * res4).sum(1)       # bases in inexact residuals
for name, key in (('inexact sides x 64 + total residual', nin * 64 + np.minimum(tot, 63)), ('bases in inexact residuals', inlen),
                  ('inexact bases x 4 + total residual', inlen * 4 + tot), ('inexact bases x 4 + total + 30 x chain-pair cost', inlen * 4 + tot + 30 * pc)):
    print('   sorted by %-50s %.3f   corr with cost %.2f' % (name + ':', fill(li[np.argsort(-key[li], kind="stable")]), np.corrcoef(key[li], cost[li])[0, 1]))
X2 = np.stack([np.ones(len(li)), tot[li], mx[li], gap2[li], pc[li], (res4[li] > 0).sum(1), nin[li], inlen[li]], 1)
w, *_ = np.linalg.lstsq(X2, cost[li], rcond=None)
pred = X2 @ w
print('   linear model incl. inexact sides / bases: %.3f  corr %.2f  weights %s' % (fill(li[np.argsort(-pred, kind="stable")]), np.corrcoef(pred, cost[li])[0, 1], np.round(w, 3).tolist()))
q = np.digitize(pred, np.quantile(pred, np.linspace(0, 1, 17)[1:-1]))          # 16 levels of the model as one radix key
print('   the same model quantised to 16 levels (one 4-bit radix pass), ties in input order: %.3f' % fill(li[np.argsort(-q, kind="stable")]))
q8 = np.digitize(pred, np.quantile(pred, np.linspace(0, 1, 257)[1:-1]))
print('   256 levels (two passes): %.3f' % fill(li[np.argsort(-q8, kind="stable")]))
