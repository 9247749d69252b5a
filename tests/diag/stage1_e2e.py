"""cm_mapping_run end to end at BASELINE configs[1] scale (chr21-like contig, 1 M pairs) from files on the GPU box:
where the wall time goes once the mapping itself takes 10 ms per million pairs.
usage: python tests/diag/stage1_e2e.py [pairs] [report 0|1|2]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa: E402,F401

if torch.cuda.is_available():
    torch.cuda.init()
from circminer_amd import lib as cl, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
report = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = synth.generate("chr21", n_pairs=n, seed=21)
with tempfile.TemporaryDirectory() as td:
    fa = os.path.join(td, "ref.fa")
    with open(fa, "w") as f:
        for name, con, start, ln in d.chr_table:
            f.write(f">{name}\n")
            f.write(d.contigs[con - 1][start:start + ln].tobytes().decode())
            f.write("\n")
    t = time.time(); packed, info = cl.pack_genome(fa); print(f"pack_genome {time.time() - t:.1f}s", flush=True)
    t = time.time(); idx = cl.write_index(packed, kmer=20, n_threads=16); print(f"write_index {time.time() - t:.1f}s ({os.path.getsize(idx) / 1e9:.2f} GB)", flush=True)
    gtf = os.path.join(td, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
    L = d.seq1.shape[1]
    q = ("I" * L + "\n").encode()
    fq = []
    t = time.time()
    for mate, arr in ((1, d.seq1), (2, d.seq2)):
        p = os.path.join(td, f"r_{mate}.fq")
        with open(p, "wb") as f:
            for i in range(n):
                f.write(b"@p%d/%d\n" % (i, mate) + arr[i].tobytes() + b"\n+\n" + q)
        fq.append(p)
    print(f"fastq written {time.time() - t:.1f}s ({os.path.getsize(fq[0]) * 2 / 1e6:.0f} MB)", flush=True)
    for it in range(2):
        t = time.time()
        st = cl.run_mapping(idx, gtf, fq[0], fq[1], os.path.join(td, "out"), cl.default_params(kmer=0), report=report, n_threads=16, batch_pairs=int(os.environ.get("BATCH", "0")))
        wall = time.time() - t
        print(f"run {it}: wall {wall:.2f}s = load {st.seconds_load:.2f}s (index file + GTF -> HBM) + map {st.seconds_map:.2f}s "
              f"({st.pairs / st.seconds_map / 1e6:.2f} M pairs/s from FASTQ text to {'none PAM SAM'.split()[report]} rows); "
              f"parse {st.seconds_parse:.2f}s, device {st.seconds_device:.2f}s, write {st.seconds_write:.2f}s (overlapped); "
              f"{st.bsj_pairs} BSJ pairs, types {list(st.by_type)}", flush=True)
