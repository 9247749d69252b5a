"""Per-step wall time of the default bench workload (diagnostic, GPU box): warm-up curve."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import dist as cdist, lib as cl, synth
if os.environ.get("WITH_TORCH"):
    import torch
    torch.zeros(1, device="cuda").add_(1); torch.cuda.synchronize()
d = synth.generate("chr21", n_pairs=1_000_000, seed=21, read_seed=0)
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=16)
P = cl.default_params(device=0); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0]); hp.upload(cl.ReadBatch(d.seq1, d.seq2))
ts = []
for it in range(14):
    if it == 4: hp.prof(True); hp.prof_reset()
    t = time.perf_counter(); hp.reset(); hp.map_round(0, True); idx, st = hp.collect_active(); rec = cdist.pack_records(idx, st); hp.sync()
    ts.append((time.perf_counter() - t) * 1e3)
print([round(x, 2) for x in ts])
