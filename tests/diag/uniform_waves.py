"""What does a k_pair wave cost when its 64 pairs are (a) the same pair 64 times, (b) 64 pairs of one category picked at random?
CM_LANE_CLK=1 wave times (product build)."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['CM_LANE_CLK'] = '1'
from circminer_amd import lib as cl, synth
N = 65536
d = synth.generate(os.environ.get('PRESET', 'chr21'), n_pairs=N, seed=21)
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs[:1], [t for t in d.chr_table if t[1] == 1], '/tmp/c.gtf', n_threads=16)
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
hp.L.cm_debug_lane_clk.argtypes = [C.c_void_p, C.c_void_p]

def run(s1, s2, label):
    b = cl.ReadBatch(s1, s2); hp.upload(b)
    hp.reset(); hp.map_round(0, True); hp.sync()
    clk = np.zeros(b.n, np.uint64); assert hp.L.cm_debug_lane_clk(hp.h, clk.ctypes.data) == 0
    st, cat, act = hp.download()
    w = (clk & np.uint64(0xFFFFFFFF)).astype(np.float64) / 100.0
    w = w[w > 0]
    wv = w[::64]
    print('%-58s waves %5d  wave time mean %7.1f us  p50 %7.1f  min %6.1f   cats %s' % (label, len(wv), wv.mean(), np.median(wv), wv.min(), np.bincount(np.clip(cat, 0, 11), minlength=12).tolist()))
    return cat

cat = run(d.seq1, d.seq2, 'the synthetic batch as it is')
exact = np.nonzero(cat == 0)[0]
for k in range(3):
    i = exact[k * 7]
    run(np.repeat(d.seq1[i:i + 1], N, 0), np.repeat(d.seq2[i:i + 1], N, 0), 'one concordant pair x %d (pair %d, src %d)' % (N, i, d.src[i]))
sel = exact[np.arange(N) % len(exact)]
run(d.seq1[sel], d.seq2[sel], 'concordant pairs only, all different')
g = exact[d.src[exact] == 1]
if len(g):
    sel = g[np.arange(N) % len(g)]
    run(d.seq1[sel], d.seq2[sel], 'concordant pairs drawn from the genome (no splice)')
t = exact[d.src[exact] == 0]
sel = t[np.arange(N) % len(t)]
run(d.seq1[sel], d.seq2[sel], 'concordant pairs drawn from transcripts')
# 64 copies of each of 1024 different pairs, arranged so that a wave holds 64 different pairs vs one pair
base = exact[:1024]
sel = np.repeat(base, 64)
run(d.seq1[sel], d.seq2[sel], '1024 pairs x 64 copies, copies adjacent in the input')
