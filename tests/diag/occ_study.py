"""k_pair time vs waves/SIMD on short reads (LDS staging small enough for 3-4 waves): is occupancy worth buying?"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import lib as cl, synth
L = int(os.environ.get("READ_LEN", "80"))
d = synth.generate("chr21", n_pairs=1_000_000, seed=21, read_len=L)
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=16)
P = cl.default_params(device=0); hp = cl.HotPath(P); hp.load_contig(0, hi.views[0], hi.annots[0]); hp.upload(cl.ReadBatch(d.seq1, d.seq2))
for it in range(8):
    if it == 4: hp.prof(True); hp.prof_reset()
    hp.reset(); hp.map_round(0, True); hp.sync()
ms, n, cnt = hp.prof_get()
print(os.environ.get("CM_LIB", "default"), "read_len", L, {k: round(v / 4, 2) for k, v in zip(["seed", "chain", "pair_stage", "scan", "heavy", "cls", "chain_heavy", "mid"], ms)})
