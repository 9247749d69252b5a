"""Stage 2 (cm_circ_call) throughput on the build box, CPU only: remain pairs per second for 1 thread and all threads.
With a -DCM_S2_PROF build (DIAG_FLAGS=-DCM_S2_PROF DIAG_NAME=s2p bash tests/diag/build_diag.sh; CM_LIB=tests/_hostemu/libcmhot_s2p.so)
the library also prints where one thread spends its time (table build / chaining / re-alignment / placement)."""
import os, sys, time, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, stage2_util as s2
from circminer_amd import lib as cl
from oracle import oracle_py as op
op.build()
tmp = pathlib.Path(tempfile.mkdtemp())
ds = conftest.DataSet(tmp, "small", int(os.environ.get("PAIRS", "40000")), 23, mix=(0.5, 0.2, 0.3))
P = cl.default_params()
st, act, cat = op.map_all_rounds(P, ds.ohi, ds.batch)
prefix, r1, r2 = s2.remain_files_from_states(tmp, ds.d, P, st, act, ds.hi.n_contigs)
print("pairs", ds.batch.n, "still active after stage 1 (back-splice candidates)", int(act.sum()))
g1, g2 = s2.gnu_sort(r1), s2.gnu_sort(r2)
rd = cl.FastqReader(g1, g2, ds.d.chr_table, P.max_ed)
b = rd.next_batch(1 << 30)
for threads in ("1", None):
    if threads:
        os.environ["CM_CIRC_THREADS"] = threads
    else:
        os.environ.pop("CM_CIRC_THREADS", None)
    t = time.time()
    stats = cl.circ_call(P, ds.hi, ds.d.chr_table, b, str(tmp / "p.cand"), str(tmp / "p.rep"))
    dt = time.time() - t
    print("threads %s: %.2f s for %d remain pairs -> %.0f pairs/s (%d candidate rows, %d calls)" % (threads or "all", dt, b.n, b.n / dt, stats.candidate_rows, stats.calls))
