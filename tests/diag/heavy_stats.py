"""How much of an hg38-like batch is 'heavy' work: chaining problems by hit count, pairs by chain-pair cost (round 1 state)."""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import lib as cl, synth
n = int(os.environ.get("PAIRS", "100000"))
preset = os.environ.get("PRESET", "hg38like")
d = synth.generate(preset, n_pairs=n, seed=38 if preset.startswith("hg38") else 21)
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=32)
P = cl.default_params(); hp = cl.HotPath(P)
b = cl.ReadBatch(d.seq1, d.seq2)
for ci in range(hi.n_contigs):
    hp.load_contig(ci, hi.views[ci], hi.annots[ci])
hp.upload(b)
for ci in range(hi.n_contigs):
    st, cat, act = hp.download()
    a, c, raw, S = hp.seeds(ci)
    c = c.reshape(n * 4, S).astype(np.int64); raw = raw.reshape(n * 4, S)
    am = np.repeat(act.astype(bool), 4)
    hits = c.sum(1)
    # (hit, later hit) pairs
    suffix = np.cumsum(c[:, ::-1], 1)[:, ::-1]
    w = (c[:, :-1] * suffix[:, 1:]).sum(1)
    heavy_chain = ((w > 256) | (hits > 96)) & am
    ch, nc, hh = hp.chains(ci)
    nc = nc.reshape(n, 4).astype(np.int64)
    cost = nc[:, 0] * nc[:, 3] + nc[:, 2] * nc[:, 1] + nc.sum(1)
    actb = act.astype(bool)
    hv = (cost > 8) & actb
    print(f"round {ci}: active pairs {actb.sum()} ({actb.mean():.3f}); chaining problems with hits {((hits > 0) & am).sum()}, heavy {heavy_chain.sum()} "
          f"(hit-pairs in heavy: {w[heavy_chain].sum():.3g} of {w[am].sum():.3g}; max hits {hits.max()}); "
          f"pairs with chains {((nc.sum(1) > 0) & actb).sum()}, heavy pairs {hv.sum()} cost pct [50,90,99,max] {np.percentile(cost[hv], [50, 90, 99]).tolist() if hv.any() else []} {cost.max()}", flush=True)
    print("   cost histogram of active pairs:", np.bincount(np.minimum(cost[actb], 40), minlength=41).tolist(), flush=True)
    both = ((nc[:, 0] + nc[:, 1]) > 0) & ((nc[:, 2] + nc[:, 3]) > 0) & actb
    cb = cost[both]
    edges = [0, 6, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 100000]
    print("   pairs with chains on both mates by cost level", [(edges[k + 1], int(((cb > edges[k]) & (cb <= edges[k + 1])).sum())) for k in range(len(edges) - 1)],
          "sum of cost beyond 6:", int(cb[cb > 6].sum()), "of", int(cb.sum()), flush=True)
    hp.map_round(ci, ci == hi.n_contigs - 1)
hp.close()
