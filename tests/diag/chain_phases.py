"""Where a heavy chaining problem's wave time goes (load + init / DP / back-tracking): a -DCM_CHAIN_DIAG build of the library
accumulates 100 MHz ticks per phase into cm_prof_counters[5..7].  python tests/diag/chain_phases.py [pairs]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import _build
so = _build.build(tag="cdiag", flags=["-DCM_CHAIN_DIAG"])
os.environ["CM_LIB"] = so
from circminer_amd import lib as cl, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
d = synth.generate("hg38like", n_pairs=n, seed=38)
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=os.cpu_count() or 8)
P = cl.default_params(); hp = cl.HotPath(P)
for ci in range(hi.n_contigs):
    hp.load_contig(ci, hi.views[ci], hi.annots[ci])
b = cl.ReadBatch(d.seq1, d.seq2)
hp.upload(b)
hp.map_rounds([0, 1, 2], True); hp.sync(); hp.reset()
hp.prof(True); hp.prof_reset()
hp.map_rounds([0, 1, 2], True); hp.sync()
ms, nl, cnt = hp.prof_get()
print("stage ms:", [round(x, 2) for x in ms], "launches", nl)
t = [c / 1e5 for c in cnt[5:8]]          # ms of wave time
print(f"k_chain_heavy wave-ms: load+init {t[0]:.0f}, DP {t[1]:.0f}, back-tracking {t[2]:.0f}  (sum {sum(t):.0f}; "
      f"3072 resident waves -> {sum(t) / 3072:.2f} ms if perfectly packed)")
hp.close()
