"""The product's host builders (host_index.cpp: range-partitioned parallel build; host_annot.cpp) against the oracle's own
(oracle/cm_oracle_build.cpp, single thread, written from the reference) on ONE FULL-SIZE packed contig of the dense hg38-like
genome (1.06 Gbp, ~20 000 genes): every array identical; digests + timings -> profiles/r03_builders_1g.json.
Pure host work (no GPU): ~30 GB of memory.   python tests/diag/builders_1g.py [threads]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from circminer_amd import _build, lib as cl, synth
from oracle import oracle_py as op
from builders_util import assert_host_views_equal
_build.build(); op.build()
nt = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8)
t = time.time()
d = synth.generate("contig1g_dense", n_pairs=1000, seed=38)
gtf = "/tmp/builders_1g.gtf"
open(gtf, "w").write(d.gtf_text)
res = {"preset": "contig1g_dense seed 38", "contig_bp": len(d.contigs[0]), "genes": len(d.genes),
       "transcripts": sum(len(g.transcripts) for g in d.genes), "generate_s": round(time.time() - t, 1)}
print(res, flush=True)
t = time.time()
hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=nt)
res["product_builders_s"] = round(time.time() - t, 1); res["product_threads"] = nt
print("product", res["product_builders_s"], flush=True)
t = time.time()
ohi = op.OracleIndex(d.contigs, d.chr_table, gtf, kmer=20)
res["oracle_builders_s"] = round(time.time() - t, 1)
print("oracle", res["oracle_builders_s"], flush=True)
dig = []
assert_host_views_equal(hi, ohi, dig)
res["identical"] = True
res["n_entries"] = int(hi.views[0].n_entries)
res["sha256"] = dig[0]
json.dump(res, open(os.path.join(ROOT, "profiles", "r03_builders_1g.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
