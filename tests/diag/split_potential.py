"""Would a fast kernel (pairs that never reach a real DP) + a slow kernel (the others) beat one kernel over the mix?
Pair-stage times of: the mix, the no-DP pairs alone, the DP pairs alone (each filled up to N pairs), on the product build.
The per-pair DP counts come from the host emulation of the same kernel bodies."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from circminer_amd import lib as cl, synth
import conftest
from oracle import oracle_py as op
N = int(os.environ.get('PAIRS', '1048576'))
M = int(os.environ.get('SAMPLE', '262144'))            # pairs the emulation classifies
d = synth.generate(os.environ.get('PRESET', 'chr21'), n_pairs=N, seed=21)
open('/tmp/c.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs[:1], [t for t in d.chr_table if t[1] == 1], '/tmp/c.gtf', n_threads=16)
P = cl.default_params(); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0])
E = conftest.load_emu(); op.build()
E.emu_set_dp_out.argtypes = [C.c_void_p]
dps = np.zeros(M, np.uint32); E.emu_set_dp_out(dps.ctypes.data)
bs = cl.ReadBatch(d.seq1[:M], d.seq2[:M])
st, act = op.default_state(P, M); cat = np.full(M, -1, np.int32)
assert E.emu_map_round(C.byref(P), C.byref(hi.views[0]), C.byref(hi.annots[0]), C.byref(bs.c), 1, st.ctypes.data, act.ctypes.data, cat.ctypes.data) == 0
E.emu_set_dp_out(None)
print('pairs with no real DP: %.1f%%; DPs per pair among the others: mean %.1f' % (100 * (dps == 0).mean(), dps[dps > 0].mean()))
for c in range(12):
    m = cat == c
    if m.sum() > 50: print('   category %2d: %6d pairs, %.1f%% without a DP' % (c, m.sum(), 100 * (dps[m] == 0).mean()))

def run(sel, label):
    sel = sel[np.arange(N) % len(sel)]
    b = cl.ReadBatch(d.seq1[sel], d.seq2[sel]); hp.upload(b)
    for rep in range(3):
        hp.reset(); hp.prof(True); hp.prof_reset(); hp.map_round(0, True); hp.sync()
        ms, n, cnt = hp.prof_get()
    print('%-40s pair stage %.2f ms (seed %.2f, chain %.2f)' % (label, ms[2], ms[0], ms[1]))
    return ms[2]

t_mix = run(np.arange(M), 'the mix')
t_fast = run(np.nonzero(dps == 0)[0], 'pairs without a DP only')
t_slow = run(np.nonzero(dps > 0)[0], 'pairs with a DP only')
f = (dps > 0).mean()
print('weighted sum %.2f ms vs mix %.2f ms' % (f * t_slow + (1 - f) * t_fast, t_mix))
