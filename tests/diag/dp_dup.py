"""How many of a pair's X-drop DP requests are repeats?  Every (mate-pair) task of process_mates extends its two chains on both
sides; tasks (i, j) and (i, j') ask for the same genomic DP of chain i when no transcript path applies.  Counts, per pair of one
mapping round on the host emulation of the kernel bodies (CPU only): requests / distinct requests, recurrences run / distinct
recurrences (a request = local_alignment_sc called; a recurrence = not answered by the exact-prefix fast path)."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from circminer_amd import lib as cl, synth
import conftest
from oracle import oracle_py as op
N = int(os.environ.get('PAIRS', '20000'))
preset = os.environ.get('PRESET', 'hg38like')
d = synth.generate(preset, n_pairs=N, seed=int(os.environ.get('SEED', '38')))
open('/tmp/dd.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs[:1], [t for t in d.chr_table if t[1] == 1], '/tmp/dd.gtf', n_threads=8)
P = cl.default_params()
b = cl.ReadBatch(d.seq1, d.seq2)
E = conftest.load_emu()
E.emu_set_dp_dup_out.argtypes = [C.c_void_p]
E.emu_set_dp_out.argtypes = [C.c_void_p]
dup = np.zeros((b.n, 4), np.uint32); E.emu_set_dp_dup_out(dup.ctypes.data)
op.build()
st, act = op.default_state(P, b.n); cat = np.full(b.n, -1, np.int32)
rc = E.emu_map_round(C.byref(P), C.byref(hi.views[0]), C.byref(hi.annots[0]), C.byref(b.c), 1, st.ctypes.data, act.ctypes.data, cat.ctypes.data)
E.emu_set_dp_dup_out(None)
print('rc', rc, 'pairs', b.n)
req, ureq, dp, udp = (dup[:, k].astype(np.int64) for k in range(4))
print('requests %d distinct %d (%.2fx);  recurrences %d distinct %d (%.2fx)' % (req.sum(), ureq.sum(), req.sum() / max(1, ureq.sum()), dp.sum(), udp.sum(), dp.sum() / max(1, udp.sum())))
edges = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 1 << 30]
for k in range(len(edges) - 1):
    m = (dp > edges[k] - (k == 0)) & (dp <= edges[k + 1]) if k else (dp == 0)
    if k == 0: m = dp == 0
    else: m = (dp > edges[k - 1] if k > 1 else dp > 0) & (dp <= edges[k])
    if m.any():
        print('  pairs with recurrences in (%d, %d]: %7d   recurrences %8d distinct %8d   requests %8d distinct %8d' % (edges[k - 1] if k else -1, edges[k], m.sum(), dp[m].sum(), udp[m].sum(), req[m].sum(), ureq[m].sum()))
