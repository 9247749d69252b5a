"""hg38-sized synthetic genome (24 chromosomes, 3.09 Gbp -> 3 packed contigs = 3 rounds, all resident in HBM): build, map
1 M pairs through all rounds, check a slice against the oracle.  Scale evidence for BASELINE.json configs[2..4]; needs
~60 GB of host memory and a few minutes of host time."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import lib as cl, synth
from oracle import oracle_py as op
op.build()
n = int(os.environ.get("PAIRS", "1000000"))
t = time.time()
d = synth.generate("hg38like", n_pairs=n, seed=38)
print("generated: contigs", [len(c) for c in d.contigs], "chromosomes", len(d.chr_table), "genes", len(d.genes), "%.0fs" % (time.time() - t), flush=True)
t = time.time()
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=16)
print("index + annotation built: entries", [int(v.n_entries) for v in hi.views], "%.0fs" % (time.time() - t), flush=True)
P = cl.default_params(device=0); hp = cl.HotPath(P)
t = time.time()
for ci in range(hi.n_contigs):
    hp.load_contig(ci, hi.views[ci], hi.annots[ci])
print("loaded to HBM %.1fs" % (time.time() - t), flush=True)
b = cl.ReadBatch(d.seq1, d.seq2); hp.upload(b)
ts, per_round = [], []
for it in range(6):
    if it == 3: hp.prof(True); hp.prof_reset()
    t = time.perf_counter(); hp.reset(); rr = []
    for ci in range(hi.n_contigs):
        t1 = time.perf_counter(); hp.map_round(ci, ci == hi.n_contigs - 1); hp.sync(); rr.append((time.perf_counter() - t1) * 1e3)
    ts.append((time.perf_counter() - t) * 1e3); per_round = rr
ms, nl, cnt = hp.prof_get()
print("step ms", [round(x, 1) for x in ts], "rounds of the last step", [round(x, 1) for x in per_round],
      {k: round(v / 3, 2) for k, v in zip(["seed", "chain", "pair_stage", "scan", "heavy", "cls", "chain_heavy"], ms)}, flush=True)
print("=> %.1f M pairs/s through all %d rounds" % (n / (np.median(ts[1:]) * 1e-3) / 1e6, hi.n_contigs), flush=True)
st1, cat1, act1 = hp.download()
print("types", np.bincount(st1["type"], minlength=14).tolist(), "active after the last round", int(act1.sum()), flush=True)
N0 = int(os.environ.get("PARITY_N", "20000"))
st0, act0 = op.default_state(P, b.n)
for ci in range(hi.n_contigs):
    op.map_round(P, hi.views[ci], hi.annots[ci], b, ci == hi.n_contigs - 1, st0, act0, 0, N0)
ok = (act0[:N0] == act1[:N0]).all() and st0[:N0].tobytes() == st1[:N0].tobytes()
print("parity on the first %d pairs, all rounds:" % N0, "bit-exact" if ok else "MISMATCH", flush=True)
hp.close()
