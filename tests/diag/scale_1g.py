"""One full-size packed contig (1.06 Gbp, the size of one hg38 round): build, map 1 M pairs, check a slice against
the oracle.  Diagnostic / scale evidence; needs ~40 GB of host memory and a few minutes of host time."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import lib as cl, synth
from oracle import oracle_py as op
op.build()
t = time.time()
d = synth.generate("contig1g", n_pairs=1_000_000, seed=31)
print("generated: contigs", [len(c) for c in d.contigs], "genes", len(d.genes), "%.0fs" % (time.time() - t), flush=True)
t = time.time()
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=16)
print("index + annotation built: %d entries, %.0fs" % (hi.views[0].n_entries, time.time() - t), flush=True)
P = cl.default_params(device=0); hp = cl.HotPath(P)
t = time.time(); hp.load_contig(0, hi.views[0], hi.annots[0]); print("loaded to HBM %.1fs" % (time.time() - t), flush=True)
b = cl.ReadBatch(d.seq1, d.seq2); hp.upload(b)
ts = []
for it in range(6):
    if it == 3: hp.prof(True); hp.prof_reset()
    t = time.perf_counter(); hp.reset(); hp.map_round(0, True); hp.sync(); ts.append((time.perf_counter() - t) * 1e3)
ms, n, cnt = hp.prof_get()
print("step ms", [round(x, 1) for x in ts], {k: round(v / 3, 2) for k, v in zip(["seed", "chain", "pair_stage", "scan", "heavy", "cls", "chain_heavy"], ms)}, flush=True)
st1, cat1, act1 = hp.download()
print("types", np.bincount(st1["type"], minlength=14).tolist(), flush=True)
# parity on a slice (the oracle is single-threaded: ~70 k pairs/s)
N0 = int(os.environ.get("PARITY_N", "20000"))
st0, act0 = op.default_state(P, b.n)
cat0 = op.map_round(P, hi.views[0], hi.annots[0], b, True, st0, act0, 0, N0)
ok = (cat0[:N0] == cat1[:N0]).all() and (act0[:N0] == act1[:N0]).all() and st0[:N0].tobytes() == st1[:N0].tobytes()
print("parity on the first %d pairs:" % N0, "bit-exact" if ok else "MISMATCH", flush=True)
hp.close()
