"""Hit multiplicity of a synthetic preset's index (SURVEY 8(d) targets: ~10 % of the 20-mers with > 1 hit, ~1 % beyond seedLim).
python tests/diag/repeat_stats.py [preset] [threads]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from circminer_amd import _build, lib as cl, synth
_build.build()
preset = sys.argv[1] if len(sys.argv) > 1 else "hg38like"
nt = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 8)
t = time.time()
d = synth.generate(preset, n_pairs=1000, seed=38)
print(f"{preset}: generated in {time.time() - t:.0f}s, contigs {[len(c) for c in d.contigs]}, {len(d.genes)} genes, "
      f"{sum(len(g.transcripts) for g in d.genes)} transcripts", flush=True)
with tempfile.TemporaryDirectory() as td:
    gtf = os.path.join(td, "a.gtf")
    open(gtf, "w").write(d.gtf_text)
    t = time.time()
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=nt)
    print(f"index in {time.time() - t:.0f}s", flush=True)
t = time.time()
for ci, (n, multi, over, distinct) in enumerate(hi.hit_stats(500, nt)):
    print(f"contig {ci}: {n} indexed 20-mers, {multi / n:.4f} with > 1 hit, {over / n:.4f} beyond seedLim 500, {distinct} distinct", flush=True)
print(f"stats in {time.time() - t:.0f}s")
