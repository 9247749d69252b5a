"""Reference-window bytes the algorithm itself asks for per pair-round (pac2char calls x length, counted by the host emulation of
the kernel bodies), against SURVEY 8(d)'s formula figure of 4 x 170 B per pair-round: on the dense hg38-like workload the pairs
from repeat families extend tens of chains each.  CPU only (one packed contig of the dense preset, a sample of pairs).
python tests/diag/window_bytes.py [pairs]"""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from circminer_amd import _build, lib as cl, synth
from conftest import load_emu
_build.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = synth.generate("contig1g_dense", n_pairs=n, seed=38)
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "a.gtf"); open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=os.cpu_count() or 8)
E = load_emu()
stats = (C.c_ulonglong * 16).in_dll(E, "cm_stats")
P = cl.default_params()
b = cl.ReadBatch(d.seq1, d.seq2)
st = np.zeros(n, dtype=cl.MAPPED_DTYPE); act = np.ones(n, np.uint8); cat = np.zeros(n, np.int32)
from oracle import oracle_py as op
st, act = op.default_state(P, n)
for k in range(16): stats[k] = 0
rc = E.emu_map_round(C.byref(P), C.byref(hi.views[0]), C.byref(hi.annots[0]), C.byref(b.c), 1, st.ctypes.data, act.ctypes.data, cat.ctypes.data)
print(f"{n} pairs, one round on a dense 1.06-Gbp contig: {stats[15] / n:.0f} reference bytes requested per pair-round "
      f"(formula: 1360), {stats[8] / n:.2f} DPs per pair, {stats[3] / n:.2f} extend_side calls per pair; rc {rc}")
