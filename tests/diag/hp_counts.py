"""Sizes of the heavy-pair pipeline's lists per attempt (a -DCM_HP_DIAG build).  python tests/diag/hp_counts.py [pairs]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import _build
so = _build.build(tag="hpdiag", flags=["-DCM_HP_DIAG"])
os.environ["CM_LIB"] = so
from circminer_amd import lib as cl, synth
import ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
d = synth.generate("hg38like", n_pairs=n, seed=38)
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=os.cpu_count() or 8)
P = cl.default_params(); hp = cl.HotPath(P)
for ci in range(hi.n_contigs):
    hp.load_contig(ci, hi.views[ci], hi.annots[ci])
b = cl.ReadBatch(d.seq1, d.seq2)
hp.upload(b)
hp.prof(True); hp.prof_reset()
hp.map_rounds([0, 1, 2], True); hp.sync()
raw = (C.c_ulonglong * 32)()
hp.L.cm_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
hp.L.cm_debug_counters(hp.h, raw)
for a in range(2):
    o = [raw[8 + 8 * a + k] for k in range(6)]
    print(f"attempt {a}: pairs {o[0]}, tasks {o[1]}, unpaired chains {o[2]}, DP requests {o[3]} + {o[4]}, fall-backs (cumulative) {o[5]}   (sums over {n} pairs x 3 rounds)")
print(f"unpaired chains there are {raw[24]} of {raw[25]} chains; extended: see 'unpaired chains' above")
print(f"tasks a sequential loop would run: {raw[26]}; pairs that end inside the loop (CONCRD): {raw[27]}, at their first task: {raw[28]}")
hp.close()
