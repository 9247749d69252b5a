"""cm_mapping_run at a few million pairs with report 0 / 1 / 2 on the same files (chr21-like contig): the remain files must be the
same bytes, the type histograms equal (report 0 takes them from the device, the others from the downloaded states), and the rates
are printed.  python tests/diag/e2e_reports.py [pairs]"""
import os, sys, time, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from circminer_amd import _build, lib as cl, synth
_build.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
d = synth.generate("chr21", n_pairs=min(n, 1 << 21), seed=21)
base = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    packed = os.path.join(base, "ref.fa.packed.fa")
    with open(packed, "wb") as f:
        for ci, c in enumerate(d.contigs):
            f.write(b">%d\n" % (ci + 1)); np.ascontiguousarray(c).tofile(f); f.write(b"\n")
    with open(packed + ".index.info", "w") as f:
        for name, con, start, ln in d.chr_table:
            f.write(f"{con}\t{start}\t{start + ln}\t{name}\n")
    gtf = os.path.join(base, "ref.gtf"); open(gtf, "w").write(d.gtf_text)
    fq = [os.path.join(base, f"r_{m}.fq") for m in (1, 2)]
    have = d.seq1.shape[0]
    for m, arr in ((0, d.seq1), (1, d.seq2)):
        open(fq[m], "wb").close()
        for a in range(0, n, have):
            bench.write_fastq_fixed(fq[m], arr[:min(have, n - a)], m + 1, first=a, append=True)
    idx = cl.write_index(packed, kmer=20, n_threads=os.cpu_count() or 8)
    res = {}
    for report in [int(x) for x in os.environ.get('REPORTS', '0,1,2').split(',')]:
        out = os.path.join(base, f"rep{report}")
        t = time.time()
        st = cl.run_mapping(idx, gtf, fq[0], fq[1], out, cl.default_params(kmer=0), report=report, n_threads=os.cpu_count() or 8, batch_pairs=1 << 20)
        dt = time.time() - t
        rem = [open(f"{out}_{st.rounds}_remain_R{m}.fastq", "rb").read() for m in (1, 2)]
        res[report] = (list(st.by_type), rem)
        print(f"report {report}: {st.pairs} pairs, load {st.seconds_load:.2f}s, map {st.seconds_map:.2f}s = {st.pairs / st.seconds_map / 1e6:.2f} M pairs/s "
              f"(parse {st.seconds_parse:.2f} | device {st.seconds_device:.2f} | write {st.seconds_write:.2f}), total {dt:.2f}s, bsj {st.bsj_pairs}", flush=True)
    if len(res) == 3:
        assert res[0][0] == res[1][0] == res[2][0], "type histograms differ"
        assert res[0][1] == res[1][1] == res[2][1], "remain files differ"
        rows = sum(1 for _ in open(os.path.join(base, "rep1.mapping.pam")))
        assert rows == n, rows
        print("histograms and remain files identical across report modes;", rows, "PAM rows")
finally:
    shutil.rmtree(base, ignore_errors=True)
