"""Device-memory leak check: free HBM before / after many create -> load -> upload -> map -> collect -> destroy cycles."""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
torch.cuda.init()
from circminer_amd import lib as cl, synth
d = synth.generate("tiny2r", n_pairs=3000, seed=3)
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "a.gtf"); open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20)
P = cl.default_params()
b = cl.ReadBatch(d.seq1, d.seq2)
half = cl.ReadBatch(d.seq1[:1000], d.seq2[:1000])
free = []
for it in range(31):
    hp = cl.HotPath(P)
    for ci in range(hi.n_contigs):
        hp.load_contig(ci, hi.views[ci], hi.annots[ci])
    for batch in (b, half, b):
        hp.upload(batch)
        for ci in range(hi.n_contigs):
            hp.map_round(ci, ci == hi.n_contigs - 1)
        hp.collect_records(0); hp.collect_active(); hp.download()
    hp.load_contig(0, hi.views[1], hi.annots[1])          # slot re-load
    hp.close()
    torch.cuda.synchronize()
    free.append(torch.cuda.mem_get_info()[0])
print("free HBM after cycle 1, 11, 21, 31 (MiB):", [free[i] >> 20 for i in (0, 10, 20, 30)])
print("leak per cycle (KiB):", (free[0] - free[30]) / 30 / 1024)
