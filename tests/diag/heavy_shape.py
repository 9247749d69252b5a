"""Shape of the pair stage's work per pair, by the cost key of the light / heavy split (host emulation of the kernel bodies, CPU):
process_mates calls, pairing-predicate evaluations, mate-pair tasks (extend_task), unpaired-chain extensions, X-drop recurrences."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from circminer_amd import lib as cl, synth
import conftest
from oracle import oracle_py as op
N = int(os.environ.get('PAIRS', '20000'))
preset = os.environ.get('PRESET', 'hg38like')
d = synth.generate(preset, n_pairs=N, seed=int(os.environ.get('SEED', '38')))
open('/tmp/hs.gtf', 'w').write(d.gtf_text)
hi = cl.HostIndex(d.contigs[:1], [t for t in d.chr_table if t[1] == 1], '/tmp/hs.gtf', n_threads=8)
P = cl.default_params()
b = cl.ReadBatch(d.seq1, d.seq2)
E = conftest.load_emu()
E.emu_set_stats_out.argtypes = [C.c_void_p]
E.emu_set_dp_out.argtypes = [C.c_void_p]
stats = np.zeros((b.n, 16), np.uint64); E.emu_set_stats_out(stats.ctypes.data)
nchain = np.zeros(b.n * 4, np.int32)
op.build()
st, act = op.default_state(P, b.n); cat = np.full(b.n, -1, np.int32)
rc = E.emu_map_round(C.byref(P), C.byref(hi.views[0]), C.byref(hi.annots[0]), C.byref(b.c), 1, st.ctypes.data, act.ctypes.data, cat.ctypes.data)
E.emu_set_stats_out(None)
s = stats.astype(np.int64)
tasks, T, unp, calls = s[:, 0], s[:, 1], s[:, 2], s[:, 6]
dps = s[:, 11:15].sum(1)
print('rc', rc, 'pairs', b.n, 'with process_mates calls', (calls > 0).sum())
edges = [0, 6, 12, 24, 48, 96, 192, 384, 768, 2000]
print('by predicate evaluations T of the pair (both attempts):')
for k in range(len(edges) - 1):
    m = (T > edges[k]) & (T <= edges[k + 1])
    if m.any():
        print('  T in (%4d, %4d]: pairs %6d  calls/pair %.2f  tasks/pair %6.2f (max %3d)  unpaired ext/pair %6.2f  recurrences/pair %6.2f   share of all recurrences %.3f' % (
            edges[k], edges[k + 1], m.sum(), calls[m].mean(), tasks[m].mean(), tasks[m].max(), unp[m].mean(), dps[m].mean(), dps[m].sum() / max(1, dps.sum())))
m = T > 48
print('tasks per process_mates call, pairs with T > 48: histogram of tasks/pair', np.bincount(np.minimum(tasks[m], 40)).tolist())
print('unpaired extensions per pair, T > 48:', np.bincount(np.minimum(unp[m], 61)).tolist())
