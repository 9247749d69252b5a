"""stage-2 back-splice-junction calling of the product (circminer_amd/csrc/host_circ_call.cpp, host code) against
the oracle's restatement of ProcessCirc (oracle/cm_oracle.cpp), byte for byte on <out>.candidates.pam and <out>.circ_report, and
against the planted truth of the synthetic generator.  Stage-1 states come from the oracle here (no GPU on this box); the GPU
suite repeats the comparison on remain files written by the device path (tests/test_gpu_parity.py)."""
import hashlib
import json
import os
import shutil

import numpy as np
import pytest

from circminer_amd import lib as cl, synth
from oracle import oracle_py as op
from stage2_util import gnu_sort, oracle_stage2, remain_files_from_states

pytestmark = pytest.mark.skipif(not all(shutil.which(x) for x in ("paste", "sort", "tr")), reason="GNU coreutils not on PATH")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stage2_tiny_seed5.json")


def _case(tmp, preset, n, seed, **kw):
    d = synth.generate(preset, n_pairs=n, seed=seed, mix=(0.3, 0.1, 0.6))
    gtf = os.path.join(str(tmp), "a.gtf")
    open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf)                    # the product's builders: what the product's stage 2 runs on
    ohi = op.OracleIndex(d.contigs, d.chr_table, gtf)                 # the oracle's own: what the oracle's stage 2 runs on
    P = cl.default_params(**kw)
    st, act, _ = op.map_all_rounds(P, ohi, cl.ReadBatch(d.seq1, d.seq2))
    prefix, r1, r2 = remain_files_from_states(tmp, d, P, st, act, hi.n_contigs)
    return d, gtf, (hi, ohi), P, prefix, r1, r2


@pytest.mark.parametrize("preset,n,seed,kw", [("tiny", 3000, 5, {}), ("tiny2r", 3000, 7, {}), ("small", 12000, 9, {}),
                                              ("tiny", 3000, 11, dict(max_ed=6, max_sc=10)), ("small", 12000, 17, dict(max_ed=2)),
                                              ("tiny2r", 3000, 13, dict(scan_level=2, max_ed=8, seed_lim=1000)), ("tiny", 2500, 19, dict(band=2)),
                                              ("variety", 4000, 31, {}), ("variety", 4000, 32, dict(max_ed=6))])
def test_stage2_outputs_equal_the_oracle(built, tmp_path, preset, n, seed, kw):
    d, gtf, (hi, ohi), P, prefix, r1, r2 = _case(tmp_path, preset, n, seed, **kw)
    s1, s2 = cl.sort_remain(r1), cl.sort_remain(r2)                       # the product's own sort ...
    assert open(s1, "rb").read() == open(gnu_sort(r1), "rb").read()       # ... is GNU sort's order
    want_c, want_r = oracle_stage2(tmp_path, ohi, d, P, r1 + ".gnu", gnu_sort(r2))
    rd = cl.FastqReader(s1, s2, d.chr_table, P.max_ed)
    b = rd.next_batch(1 << 30)
    st = cl.circ_call(P, hi, d.chr_table, b, prefix + ".candidates.pam", prefix + ".circ_report")
    rd.close()
    got_c, got_r = open(prefix + ".candidates.pam", "rb").read(), open(prefix + ".circ_report", "rb").read()
    assert got_c == want_c and got_r == want_r
    assert st.candidate_rows == want_c.count(b"\n") > 100 and want_r.count(b"\n") > 20
    # planted truth: every reported circle is a planted back-splice at its exact coordinates, consensus = reference ("Pass"),
    # and most planted circles with enough support are found
    planted = {}
    for i in np.nonzero(d.src == 2)[0]:
        key = (d.chr_names[d.truth_chr[i]], int(d.truth_lo[i]), int(d.truth_hi[i]))
        planted[key] = planted.get(key, 0) + 1
    rows = [r.split("\t") for r in got_r.decode().strip().split("\n")]
    found = {(r[0], int(r[1]), int(r[2])) for r in rows}
    if preset != "variety":          # (overlapping genes of the variety preset also yield circles the generator did not plant as such)
        assert found <= set(planted), sorted(found - set(planted))[:5]
        assert len(found) >= (0.8 if P.max_ed >= 4 else 0.6) * len(planted)
    assert sum(r[7] == "Pass" for r in rows) >= 0.95 * len(rows) and all(r[4] == "STC" for r in rows)
    if (preset, seed) == ("tiny", 5) and not kw:                          # regression guard (digests of the oracle's own output)
        g = json.load(open(GOLDEN))
        assert hashlib.sha256(want_c).hexdigest() == g["candidates_sha256"] and hashlib.sha256(want_r).hexdigest() == g["report_sha256"]
        assert want_r.decode().split("\n")[:3] == g["first_report_rows"]


def test_stage2_from_files_to_files(built, tmp_path):
    """cm_circ_run = circ_detect() of the reference (src/circminer.cpp:347-352): remain files + index file + GTF in,
    candidates.pam + circ_report out; two packed contigs (the genome is reloaded when the contig changes)."""
    d, gtf, (hi, ohi), P, prefix, r1, r2 = _case(tmp_path, "tiny2r", 2500, 23)
    fa = str(tmp_path / "ref.fa")
    with open(fa, "w") as f:
        for name, con, start, ln in d.chr_table:
            f.write(f">{name}\n{d.contigs[con - 1][start:start + ln].tobytes().decode()}\n")
    packed, info = cl.pack_genome(fa, 150_000)
    idx = cl.write_index(packed, kmer=20, n_threads=4)
    st = cl.run_circ(idx, gtf, prefix, hi.n_contigs, cl.default_params(kmer=0))
    want_c, want_r = oracle_stage2(tmp_path, ohi, d, P, gnu_sort(r1), gnu_sort(r2))
    assert open(prefix + ".candidates.pam", "rb").read() == want_c and open(prefix + ".circ_report", "rb").read() == want_r
    assert st.calls > 0 and st.pairs == open(r1).read().count("\n") // 4
    contigs_seen = {r.split("\t")[1] for r in want_c.decode().strip().split("\n")}
    assert contigs_seen == {"chr1", "chr2"}
    with pytest.raises(RuntimeError):
        cl.run_circ(idx + ".nope", gtf, prefix, hi.n_contigs)


def test_stage2_empty_and_unannotated(built, tmp_path):
    """no BSJ candidates at all -> both files exist and are empty; a pair outside every gene is skipped ("Gene not found")"""
    d = synth.generate("tiny", n_pairs=300, seed=3, mix=(0.0, 1.0, 0.0))
    gtf = os.path.join(str(tmp_path), "a.gtf")
    open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf)
    P = cl.default_params()
    st, act, _ = op.map_all_rounds(P, hi, cl.ReadBatch(d.seq1, d.seq2))
    act[:] = 0
    prefix, r1, r2 = remain_files_from_states(tmp_path, d, P, st, act, 1)
    s1, s2 = cl.sort_remain(r1), cl.sort_remain(r2)
    rd = cl.FastqReader(s1, s2, d.chr_table, P.max_ed)
    assert rd.next_batch(10) is None
    rd.close()
    # a fabricated CHIBSJ state far from any gene
    st2 = st[:1].copy()
    st2["type"] = 3; st2["chr_id"] = 0; st2["contig_num"] = 0
    st2["spos_r1"] = 100; st2["epos_r1"] = 199; st2["qspos_r1"] = 1; st2["qepos_r1"] = 100; st2["mlen_r1"] = 100
    st2["spos_r2"] = 300; st2["epos_r2"] = 449; st2["qspos_r2"] = 1; st2["qepos_r2"] = 150; st2["mlen_r2"] = 150
    prefix, r1, r2 = remain_files_from_states(tmp_path, d, P, np.repeat(st2, 300), np.r_[1, np.zeros(299, np.uint8)], 1, out="q")
    rd = cl.FastqReader(cl.sort_remain(r1), cl.sort_remain(r2), d.chr_table, P.max_ed)
    b = rd.next_batch(10)
    stt = cl.circ_call(P, hi, d.chr_table, b, prefix + ".candidates.pam", prefix + ".circ_report")
    assert stt.pairs == 1 and stt.candidate_rows == 0 and open(prefix + ".circ_report").read() == ""
    rd.close()


@pytest.mark.parametrize("threads", ["1", "3", "13"])
def test_stage2_does_not_depend_on_the_thread_count(built, tmp_path, monkeypatch, threads):
    """cm_circ_call cuts the sorted pairs into chunks that worker threads take in turn (a Caller and a table cache each); rows and
    calls are assembled in input order, so the files are the same bytes for any number of threads (and equal the oracle's)."""
    monkeypatch.setenv("CM_CIRC_THREADS", threads)
    d, gtf, (hi, ohi), P, prefix, r1, r2 = _case(tmp_path, "tiny2r", 4000, 23)
    s1, s2 = cl.sort_remain(r1), cl.sort_remain(r2)
    want_c, want_r = oracle_stage2(tmp_path, ohi, d, P, gnu_sort(r1), gnu_sort(r2))
    rd = cl.FastqReader(s1, s2, d.chr_table, P.max_ed)
    b = rd.next_batch(1 << 30)
    cl.circ_call(P, hi, d.chr_table, b, prefix + ".candidates.pam", prefix + ".circ_report")
    rd.close()
    assert open(prefix + ".candidates.pam", "rb").read() == want_c and open(prefix + ".circ_report", "rb").read() == want_r
    assert want_c.count(b"\n") > 500            # more rows than one chunk holds: several chunks per contig
