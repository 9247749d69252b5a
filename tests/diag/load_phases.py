"""Where cm_mapping_run's load time goes at chr21 scale: HIP context, index record -> host arrays, host -> HBM + descriptors."""
import ctypes as C, os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
torch.cuda.init()
from circminer_amd import lib as cl, synth
d = synth.generate("chr21", n_pairs=1000, seed=21)
td = tempfile.mkdtemp()
fa = os.path.join(td, "ref.fa")
with open(fa, "w") as f:
    for name, con, start, ln in d.chr_table:
        f.write(f">{name}\n{d.contigs[con - 1][start:start + ln].tobytes().decode()}\n")
packed, info = cl.pack_genome(fa)
idx = cl.write_index(packed, kmer=20, n_threads=16)
gtf = os.path.join(td, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
for it in range(2):
    t = time.time(); hp = cl.HotPath(cl.default_params()); t_ctx = time.time() - t
    t = time.time(); f = cl.IndexFile(idx, n_threads=16); iv = next(f); t_file = time.time() - t
    t = time.time(); hp.L.cm_load_contig(hp.h, 0, C.byref(iv)); hp.sync(); t_up = time.time() - t
    print(f"run {it}: context {t_ctx:.2f}s, index record -> host arrays {t_file:.2f}s, host -> HBM + descriptors {t_up:.2f}s")
    f.close(); hp.close()
