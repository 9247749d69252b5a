"""Host-side breakdown of one bench step (diagnostic, run on the GPU box)."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import dist as cdist, lib as cl, synth

d = synth.generate("chr21", n_pairs=1_000_000, seed=21, read_seed=0)
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "ref.gtf")
    open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=16)
P = cl.default_params(device=0)
hp = cl.HotPath(P)
hp.load_contig(0, hi.views[0], hi.annots[0])
hp.upload(cl.ReadBatch(d.seq1, d.seq2))
acc = {}
def tick(name, t0):
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0)
for it in range(6):
    if it == 1:
        acc.clear()
    t = time.perf_counter(); hp.reset(); tick("reset(launch)", t)
    t = time.perf_counter(); hp.map_round(0, True); tick("map_round(launch+tile sync)", t)
    t = time.perf_counter(); hp.sync(); tick("sync", t)
    t = time.perf_counter(); idx, st = hp.collect_active(); tick("collect_active", t)
    t = time.perf_counter(); rec = cdist.pack_records(idx, st); tick("pack_records", t)
print({k: round(v / 5 * 1e3, 3) for k, v in acc.items()}, len(rec))
