"""wall time of cm_host_build_annotation on an Ensembl-scale GTF (60 k genes, ~200 k transcripts,
~1.3 M exon lines, hg38 chromosome lengths packed into contigs the way GenomePacker does).  CPU only.
usage: python tests/diag/annot_scale.py [n_genes]"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from circminer_amd import lib as cl  # noqa: E402

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
        133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895,
        57227415]
CONTIG = 1100000000


def chr_table():
    rows, con, pos = [], 1, 0
    for i, ln in enumerate(HG38):
        if pos and pos + ln > CONTIG:
            con, pos = con + 1, 0
        rows.append((f"{i + 1}", con, pos, ln))
        pos += ln + 50
    return rows


def write_gtf(path, rows, n_genes, seed=3):
    rng = np.random.default_rng(seed)
    tot = sum(HG38)
    n_lines = 0
    with open(path, "w") as f:
        for name, _con, _sh, ln in rows:
            ng = max(1, int(n_genes * ln / tot))
            starts = np.sort(rng.integers(1000, ln - 400000, ng))
            for gi, gs in enumerate(starts):
                gs = int(gs)
                strand = "+-"[int(rng.integers(0, 2))]
                n_ex = int(rng.integers(1, 25))
                cuts = np.sort(rng.choice(np.arange(1, 3000), size=2 * n_ex, replace=False)) * int(rng.integers(5, 60))
                ex = [(gs + int(cuts[2 * k]), gs + int(cuts[2 * k + 1]) - 1) for k in range(n_ex)]
                ge = ex[-1][1]
                gid = f"G{name}_{gi}"
                f.write(f'{name}\tsynth\tgene\t{ex[0][0]}\t{ge}\t.\t{strand}\t.\tgene_id "{gid}"; gene_name "{gid}";\n')
                n_lines += 1
                for ti in range(int(rng.integers(1, 7))):
                    keep = [e for e in ex if rng.random() < 0.8] or ex[:1]
                    f.write(f'{name}\tsynth\ttranscript\t{keep[0][0]}\t{keep[-1][1]}\t.\t{strand}\t.\tgene_id "{gid}"; transcript_id "{gid}.{ti}"; gene_name "{gid}";\n')
                    for k, (a, b) in enumerate(keep):
                        f.write(f'{name}\tsynth\texon\t{a}\t{b}\t.\t{strand}\t.\tgene_id "{gid}"; transcript_id "{gid}.{ti}"; exon_number "{k + 1}"; gene_name "{gid}";\n')
                    n_lines += 1 + len(keep)
    return n_lines


if __name__ == "__main__":
    n_genes = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    rows = chr_table()
    n_con = max(r[1] for r in rows)
    con_len = [0] * n_con
    for _n, con, sh, ln in rows:
        con_len[con - 1] = max(con_len[con - 1], sh + ln)
    with tempfile.TemporaryDirectory() as td:
        gtf = os.path.join(td, "big.gtf")
        t = time.time()
        n_lines = write_gtf(gtf, rows, n_genes)
        print(f"gtf: {n_lines} lines, {os.path.getsize(gtf) / 1e6:.0f} MB ({time.time() - t:.0f}s to write)", flush=True)
        L = cl.load()
        chrs = cl.chr_array(rows)
        lens = (C.c_uint32 * n_con)(*con_len)
        names = [r[0].encode() for r in rows]  # keep alive
        out = (cl.AnnotView * n_con)()
        t = time.time()
        rc = L.cm_host_build_annotation(gtf.encode(), chrs, len(rows), lens, n_con, 300, out)
        dt = time.time() - t
        assert rc == 0, rc
        for c in range(n_con):
            print(f"contig {c + 1}: {out[c].n_gene} genes, {out[c].n_trans} transcripts, {out[c].n_seg} segments, {out[c].n_iv} intervals")
        print(f"cm_host_build_annotation: {dt:.2f} s")
        L.cm_host_free_annotation(out, n_con)
