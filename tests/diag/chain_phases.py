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
import ctypes as C
raw = (C.c_ulonglong * 32)()
hp.L.cm_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
hp.L.cm_debug_counters(hp.h, raw)
tk = [raw[8 + k] for k in range(5)]
wv = [raw[16 + k] / 1e5 for k in range(4)]
print(f"DP, lane time (ms, summed over lanes): binary searches {tk[0] / 1e5:.0f}, upper_bound {tk[1] / 1e5:.0f}, window loops {tk[2] / 1e5:.0f}; "
      f"{tk[3]} cells, {tk[4]} pair evaluations ({tk[4] / max(tk[3], 1):.2f} per cell)")
print(f"DP, wave time (ms): first evaluation {wv[0]:.0f}, scan + store + second evaluation {wv[2]:.0f}, barriers {wv[3]:.0f}")
print(f"back-tracking: {raw[22]} problems with a log, {raw[20] / max(raw[22], 1):.0f} events and {raw[24] / max(raw[22], 1):.0f} cells per problem, "
      f"{raw[21] / max(raw[22], 1):.1f} score levels per problem, {raw[23] / max(raw[22], 1):.0f} 64-event passes per problem")
hp.close()
