"""FASTQ text -> circ_report on the GPU box: cm_mapping_run (stage 1, remain files) then cm_circ_run (sort + stage 2), chr21-like
contig.  usage: python tests/diag/e2e_full.py [pairs]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa: E402,F401
if torch.cuda.is_available():
    torch.cuda.init()
from circminer_amd import lib as cl, synth  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
d = synth.generate("chr21", n_pairs=n, seed=21)
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    fa = os.path.join(td, "ref.fa")
    with open(fa, "w") as f:
        for name, con, start, ln in d.chr_table:
            f.write(f">{name}\n{d.contigs[con - 1][start:start + ln].tobytes().decode()}\n")
    packed, info = cl.pack_genome(fa)
    idx = cl.write_index(packed, kmer=20, n_threads=16)
    gtf = os.path.join(td, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
    L = d.seq1.shape[1]
    q = ("I" * L + "\n").encode()
    fq = []
    for mate, arr in ((1, d.seq1), (2, d.seq2)):
        p = os.path.join(td, f"r_{mate}.fq")
        with open(p, "wb") as f:
            for i in range(n):
                f.write(b"@p%d/%d\n" % (i, mate) + arr[i].tobytes() + b"\n+\n" + q)
        fq.append(p)
    out = os.path.join(td, "out")
    for it in range(2):
        t0 = time.time()
        st = cl.run_mapping(idx, gtf, fq[0], fq[1], out, cl.default_params(kmer=0), report=0, n_threads=16)
        t1 = time.time()
        cs = cl.run_circ(idx, gtf, out, st.rounds, cl.default_params(kmer=0), n_threads=64)
        t2 = time.time()
        rep = sum(1 for _ in open(out + ".circ_report"))
        print(f"run {it}: {n} pairs: stage 1 {t1 - t0:.2f}s (load {st.seconds_load:.2f}s + map {st.seconds_map:.2f}s), {st.bsj_pairs} candidates; "
              f"stage 2 {t2 - t1:.2f}s (calling {cs.seconds:.2f}s); {rep} circRNAs reported; FASTQ -> circ_report {n / (t2 - t0) / 1e6:.2f} M pairs/s", flush=True)
