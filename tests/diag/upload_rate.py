"""PCIe-inclusive rate of the boundary: per batch of host-resident reads = cm_reads_upload + all rounds +
cm_collect_records (BSJ hand-off) + cm_reads_download (states for the PAM / remain writers).
usage: python tests/diag/upload_rate.py [pairs] [batches]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa: E402,F401  (bundled HIP runtime first)

if torch.cuda.is_available():
    torch.cuda.init()
from circminer_amd import lib as cl, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = synth.generate("chr21", n_pairs=n, seed=21)
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "ref.gtf")
    open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=16)
P = cl.default_params()
hp = cl.HotPath(P)
hp.load_contig(0, hi.views[0], hi.annots[0])
pinned = os.environ.get("CM_PINNED")
if pinned:                       # reads and result arrays in page-locked memory from cm_host_alloc
    s1 = hp.host_array(d.seq1.size, np.uint8).reshape(d.seq1.shape)
    s2 = hp.host_array(d.seq2.size, np.uint8).reshape(d.seq2.shape)
    s1[:], s2[:] = d.seq1, d.seq2
    batch = cl.ReadBatch(s1, s2)
    assert batch.seq1.ctypes.data == s1.ctypes.data
    out_st, out_cat, out_act = hp.host_array(n, cl.MAPPED_DTYPE), hp.host_array(n, np.int32), hp.host_array(n, np.uint8)
else:
    batch = cl.ReadBatch(d.seq1, d.seq2)
    out_st, out_cat, out_act = np.zeros(n, cl.MAPPED_DTYPE), np.zeros(n, np.int32), np.zeros(n, np.uint8)
    out_st[:], out_cat[:], out_act[:] = out_st, 0, 0          # touch the pages once: first-touch faults are not the library's
T = {"upload": [], "rounds": [], "records": [], "download": [], "total": []}
for it in range(nb):
    t0 = time.perf_counter()
    hp.upload(batch)
    hp.sync()
    t1 = time.perf_counter()
    hp.map_round(0, True)
    hp.sync()
    t2 = time.perf_counter()
    rec = hp.collect_records(0)
    t3 = time.perf_counter()
    hp._chk(hp.L.cm_reads_download(hp.h, out_st.ctypes.data, out_cat.ctypes.data, out_act.ctypes.data), "cm_reads_download")
    t4 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
        T[k].append(v * 1e3)
for k, v in T.items():
    print(f"{k:9s} ms: first {v[0]:8.2f}  median of rest {np.median(v[1:]):8.2f}")
print(f"PCIe-inclusive: {n / (np.median(T['total'][1:]) * 1e-3) / 1e6:.1f} M pairs/s ({len(rec)} BSJ records, pinned={bool(pinned)})")
hp.close()
