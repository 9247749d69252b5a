"""Wave-level section accounting of k_pair (diag build): wave time per section and average active lanes."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['CM_LIB'] = os.path.join(ROOT, 'tests/_hostemu/libcmhot_diag.so'); os.environ['CM_LANE_CLK'] = '1'
from circminer_amd import lib as cl, synth
N = int(os.environ.get('PAIRS', '262144'))
d = synth.generate(os.environ.get('PRESET', 'chr21'), n_pairs=N, seed=int(os.environ.get('SEED', '21')))
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs[:1], [t for t in d.chr_table if t[1] == 1], '/tmp/c.gtf', n_threads=32)
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
hp.L.cm_debug_lane_clk.argtypes = [C.c_void_p, C.c_void_p]
b = cl.ReadBatch(d.seq1, d.seq2); hp.upload(b)
hp.reset(); hp.map_round(0, True); hp.sync()
if os.environ.get('ONLY_CAT'):             # a batch of one category only (e.g. 0 = concordant), all pairs different
    cat = hp.download()[1]
    sel = np.nonzero(cat == int(os.environ['ONLY_CAT']))[0]
    if os.environ.get('ONLY_SRC'): sel = sel[d.src[sel] == int(os.environ['ONLY_SRC'])]
    if os.environ.get('SAME'): sel = sel[int(os.environ['SAME']):int(os.environ['SAME']) + 1]
    sel = sel[np.arange(b.n) % len(sel)]
    b = cl.ReadBatch(d.seq1[sel], d.seq2[sel]); hp.upload(b)
    hp.reset(); hp.map_round(0, True); hp.sync()
clk = np.zeros(b.n * 33, np.uint64); assert hp.L.cm_debug_lane_clk(hp.h, clk.ctypes.data) == 0
nw = b.n // 64 + 1
w = clk[b.n * 16: b.n * 16 + nw * 64].reshape(-1, 64).astype(np.float64)
wh = clk[b.n * 16 + nw * 64: b.n * 16 + nw * 64 + 4096 * 64].reshape(-1, 64).astype(np.float64)
w = w[w.sum(1) > 0]
wt, lt = w[:, :32].sum(0) / 100.0, w[:, 32:].sum(0) / 100.0
names = {0: 'process_mates entry (prologue)', 1: 'pass1 (pairing predicate)', 2: 'pre-ext (tids, CH copies)', 3: 'is_left + both_mates entry', 4: 'middle_ed + is_concord',
         5: 'LL extension', 6: 'RL extension', 7: 'RR extension', 8: 'LR extension', 9: 'overlaps (exon lookups)', 10: 'fold', 11: 'ext: transcript loop (rest)', 16: 'trans walk -> end_step', 17: 'end_step memo hit', 18: 'end_step extend_end (pac2char + DP)', 19: 'end_step memo_put',
         20: 'trans walk -> middle_step', 21: 'middle_step extend_middle', 22: 'extend_side entry', 24: 'sc: entry (code before the DP call)', 25: 'sc: exact -> closed form', 26: 'sc: inexact compare', 27: 'sc: staging', 28: 'sc: X-drop DP', 23: 'extend_side overlap_ind',
         12: 'ext: genomic DP', 13: 'tail (finish, stores)', 14: 'leftover extensions'}
print('waves %d, wave time total %.0f us, mean %.0f us/wave, overall active lanes %.1f' % (len(w), wt.sum(), wt.sum() / len(w), lt.sum() / wt.sum()))
for k in sorted(names):
    if wt[k] > 0: print('  %-34s share %5.1f%%   active lanes %5.1f' % (names[k], 100 * wt[k] / wt.sum(), lt[k] / wt[k]))

wh = wh[wh.sum(1) > 0]
wt, lt = wh[:, :32].sum(0) / 100.0, wh[:, 32:].sum(0) / 100.0
names.update({29: 'heavy: pass1 predicate (parallel)', 30: 'heavy: pair tasks (parallel extend_task)', 31: 'heavy: fold (lane 0)', 15: 'heavy: unpaired-chain extensions', 13: 'heavy: finish'})
print('k_pair_heavy blocks %d, block time total %.0f us, lanes at ticks %.1f' % (len(wh), wt.sum(), lt.sum() / max(wt.sum(), 1)))
for k in sorted(names):
    if wt[k] > 0: print('  %-34s share %5.1f%%   lanes at tick %5.1f' % (names[k], 100 * wt[k] / wt.sum(), lt[k] / wt[k]))
