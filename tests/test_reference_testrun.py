"""The one known answer the reference itself holds for this path: the two `output.circ_report` lines its README promises for the
figshare sample test (/root/reference/README.md:78-95; ndownloader.figshare.com/files/22423638: ref.fa, ref.gtf, R1.fq, R2.fq).
The sample is not in the reference tree and there is no network here, so this test runs only where CM_REF_TESTRUN_DIR points at
the unzipped package -- and then it is the pin the oracle lacks ("parity unpinned"): the product's own file-to-file
flow must reproduce the README's two rows byte for byte, and so must the oracle's restatement of stage 2 on the same remain files.

  README commands                                            here
  ./circminer --index -r ref.fa -k 20 --thread 4             cm_host_pack_genome + cm_host_write_index
  ./circminer -r ref.fa -g ref.gtf -1 R1.fq -2 R2.fq -o out  cm_mapping_run (stage 1, GPU) + cm_circ_run (stage 2, host)
"""
import os
import shutil

import pytest

from circminer_amd import lib as cl

# README.md:94-95, tab separated as ProcessCirc::report_events writes them (src/process_circ.cpp:1601-1608)
EXPECTED = [
    "1\t586821\t608056\t8\tSTC\tCT-AG\tCT-AG\tPass\tCirc1-12,Circ1-56,Circ1-18,Circ1-50,Circ1-110,Circ1-80,Circ1-2,Circ1-4",
    "1\t805799\t810170\t9\tSTC\tTT-TA\tTT-TA\tPass\tCirc2-32,Circ2-40,Circ2-36,Circ2-74,Circ2-76,Circ2-80,Circ2-60,Circ2-44,Circ2-2",
]
SAMPLE = os.environ.get("CM_REF_TESTRUN_DIR", "")
NEED = ("ref.fa", "ref.gtf", "R1.fq", "R2.fq")


def _have_sample():
    return bool(SAMPLE) and all(os.path.exists(os.path.join(SAMPLE, f)) for f in NEED)


def test_expected_rows_are_the_readmes():
    """(always runs) the rows above are nine tab-separated columns each and name the supporting reads the README lists"""
    for row in EXPECTED:
        f = row.split("\t")
        assert len(f) == 9 and f[0] == "1" and f[4] == "STC" and f[7] == "Pass" and int(f[3]) == len(f[8].split(","))


def _run_sample(sample_dir, work):
    """The README's two commands on the files of `sample_dir`; returns (rows of the product's circ_report, rows of the oracle's)."""
    import numpy as np
    import stage2_util as s2
    from circminer_amd import synth
    from oracle import oracle_py as op
    for f in NEED:
        shutil.copy(os.path.join(sample_dir, f), work)
    fa, gtf = os.path.join(work, "ref.fa"), os.path.join(work, "ref.gtf")
    packed, info = cl.pack_genome(fa)                                  # genome.cpp:96-167
    idx = cl.write_index(packed, kmer=20, n_threads=4)                 # HashTable.c:106-254
    out = os.path.join(work, "output")
    P = cl.default_params(kmer=20)
    st = cl.run_mapping(idx, gtf, os.path.join(work, "R1.fq"), os.path.join(work, "R2.fq"), out, P, report=0, n_threads=4, index_info=info)
    cl.run_circ(idx, gtf, out, st.rounds, cl.default_params(kmer=20), n_threads=4, index_info=info)
    got = open(out + ".circ_report").read().splitlines()
    # ... and the checker itself, end to end on its own builders: oracle stage 1 (all rounds) -> remain files -> GNU sort ->
    # oracle stage 2: with the real sample this is what pins the oracle to the reference's known answer
    names, seqs = [], []
    for ln in open(fa, "rb"):
        if ln.startswith(b">"):
            names.append(ln[1:].split()[0].decode())
            seqs.append([])
        else:
            seqs[-1].append(ln.strip().upper())
    seqs = [np.frombuffer(b"".join(x), np.uint8).copy() for x in seqs]
    for a in seqs:                                                    # loadRefGenome: anything but ACGT is N
        a[~np.isin(a, np.frombuffer(b"ACGT", np.uint8))] = ord("N")
    contigs, table = synth.pack_genome(names, seqs, cl.CM_CONTIG_SIZE)
    ohi = op.OracleIndex(contigs, table, gtf, kmer=20)
    rd = cl.FastqReader(os.path.join(work, "R1.fq"), os.path.join(work, "R2.fq"), table, P.max_ed)
    b = rd.next_batch(1 << 30)
    o_st, o_act, _ = op.map_all_rounds(P, ohi, b)
    prefix = os.path.join(work, "oracle")
    r1, r2 = f"{prefix}_{ohi.n_contigs}_remain_R1.fastq", f"{prefix}_{ohi.n_contigs}_remain_R2.fastq"
    w = cl.RecordWriter(r1, r2, table)
    w.write_remain(b, o_st, np.nonzero(o_act)[0])
    w.close()
    rd.close()

    class _D:
        chr_table = table
    _, rep = s2.oracle_stage2(work, ohi, _D, P, s2.gnu_sort(r1), s2.gnu_sort(r2))
    return got, rep.decode().splitlines()


@pytest.mark.gpu
@pytest.mark.skipif(not _have_sample(), reason="CM_REF_TESTRUN_DIR does not hold the reference's figshare sample (ref.fa, ref.gtf, R1.fq, R2.fq)")
def test_readme_test_run_reproduces_the_two_circ_report_rows(tmp_path):
    got, oracle_rows = _run_sample(SAMPLE, str(tmp_path))
    assert got == EXPECTED, "\n".join(got[:10])
    assert oracle_rows == EXPECTED, "\n".join(oracle_rows[:10])


@pytest.mark.gpu
def test_the_same_flow_on_a_synthetic_sample(tmp_path):
    """Keeps the flow above alive between the days a box holds the real sample: the same four files made from the synthetic
    `tiny` preset (a FASTA with 60-base lines, a GTF, two FASTQ files) through the same function; product == oracle, circles found."""
    import stage2_util as s2
    from circminer_amd import synth
    d = synth.generate("tiny", n_pairs=1500, seed=5)
    src = tmp_path / "sample"
    src.mkdir()
    with open(src / "ref.fa", "w") as f:
        for n, sq in zip(d.chr_names, d.chr_seqs):
            t = sq.tobytes().decode()
            f.write(f">{n} synthetic\n" + "".join(t[i:i + 60] + "\n" for i in range(0, len(t), 60)))
    (src / "ref.gtf").write_text(d.gtf_text)
    p1, p2 = s2.write_fastq_pair(src, d, 1500)
    os.rename(p1, src / "R1.fq")
    os.rename(p2, src / "R2.fq")
    work = tmp_path / "work"
    work.mkdir()
    got, oracle_rows = _run_sample(str(src), str(work))
    assert len(got) >= 5 and got == oracle_rows
