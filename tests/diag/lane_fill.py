"""How full are k_pair's waves?  (product build, CM_LANE_CLK=1.)  Each lane reports its own time for its pair; 64 consecutive
slots of the processing order are one wave iteration.  fill = sum(lane time) / (64 x longest lane) per iteration: the share
of lane-slots that had a pair in flight.  1 - fill is lost to the tail of a wave (lanes that finished early and wait)."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['CM_LANE_CLK'] = '1'
from circminer_amd import lib as cl, synth
N = int(os.environ.get('PAIRS', '262144'))
d = synth.generate(os.environ.get('PRESET', 'chr21'), n_pairs=N, seed=int(os.environ.get('SEED', '21')))
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs[:1], [t for t in d.chr_table if t[1] == 1], '/tmp/c.gtf', n_threads=16)
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
hp.L.cm_debug_lane_clk.argtypes = [C.c_void_p, C.c_void_p]
b = cl.ReadBatch(d.seq1, d.seq2); hp.upload(b)
hp.reset(); hp.map_round(0, True); hp.sync()
clk = np.zeros(b.n, np.uint64); assert hp.L.cm_debug_lane_clk(hp.h, clk.ctypes.data) == 0
own = (clk & np.uint64(0xFFFFFFFF)).astype(np.float64) / 100.0      # us
n_l = int(np.nonzero(own)[0].max()) + 1 if own.any() else 0
own = own[:n_l]
pad = (-n_l) % 64
w = np.concatenate([own, np.zeros(pad)]).reshape(-1, 64)
longest = w.max(1)
print('light pairs %d of %d, wave iterations %d' % (n_l, b.n, len(w)))
print('sum of lane times %.0f us, sum of 64 x longest lane %.0f us  ->  fill %.3f' % (w.sum(), 64 * longest.sum(), w.sum() / (64 * longest.sum())))
print('longest lane per iteration: mean %.0f us, p50 %.0f, p90 %.0f, max %.0f' % (longest.mean(), np.median(longest), np.percentile(longest, 90), longest.max()))
print('lane time: mean %.1f us, p50 %.1f, p90 %.1f, p99 %.1f' % (own.mean(), np.median(own), np.percentile(own, 90), np.percentile(own, 99)))
# by position in the processing order (the radix order puts the expensive classes first or last?)
for lo in range(0, len(w), max(1, len(w) // 10)):
    seg = w[lo: lo + max(1, len(w) // 10)]
    print('  iterations %6d..: fill %.3f  longest %.0f us  mean lane %.1f us' % (lo, seg.sum() / (64 * seg.max(1).sum()), seg.max(1).mean(), seg.mean()))
# what a perfect order would give: sort all lane times, regroup
sw = np.sort(np.concatenate([own, np.zeros(pad)])).reshape(-1, 64)
print('lane times sorted and regrouped: fill %.3f (sum of longest %.0f us vs %.0f us now)' % (sw.sum() / (64 * sw.max(1).sum()), sw.max(1).sum(), longest.sum()))
