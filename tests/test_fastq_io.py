"""FASTQ ingest, the 23-token carry-over header, PAM and remain-FASTQ writers
(circminer_amd/csrc/host_fastq.cpp) against Python restatements of the reference's formats
(src/fastq_parser.cpp:178-269, src/filter.cpp:413-455, src/output.cpp:118-333)."""
import gzip
import os

import numpy as np
import pytest

from circminer_amd import lib as cl

MAPPED = {0, 1, 2, 3, 4, 7, 5}
CHRS = [("chr1", 1, 0, 5000), ("chr2", 1, 5050, 3000), ("chrX", 2, 0, 4000)]


def _fastq(path, names, seqs, quals, opener=open):
    with opener(path, "wt") as f:
        for n, s, q in zip(names, seqs, quals):
            f.write(f"@{n}\n{s}\n+\n{q}\n")


def _rand_reads(rng, n, lo=30, hi=151):
    seqs, quals = [], []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        seqs.append("".join(rng.choice(list("ACGTNacgt"), L)))
        quals.append("".join(chr(int(x)) for x in rng.integers(35, 74, L)))
    return seqs, quals


def py_remain_header(name, m, chrs):
    """write_read_category (PE), src/filter.cpp:413-449."""
    if int(m["type"]) in MAPPED:
        cn = chrs[int(m["chr_id"])][0] if m["chr_id"] >= 0 else "-"
        shift = chrs[int(m["chr_id"])][2] if m["chr_id"] >= 0 else 0
        gspos = int(m["contig_num"]) * 1100000000 + int(m["spos_r1"]) + shift
        return (f"@{name} {gspos} {m['type']} {cn} {m['spos_r1']} {m['epos_r1']} {m['mlen_r1']} {m['qspos_r1']} {m['qepos_r1']} "
                f"{'+' if m['r1_forward'] else '-'} {m['ed_r1']} {cn} {m['spos_r2']} {m['epos_r2']} {m['mlen_r2']} {m['qspos_r2']} "
                f"{m['qepos_r2']} {'+' if m['r2_forward'] else '-'} {m['ed_r2']} {m['tlen']} {m['junc_num']} {int(bool(m['gm_compatible']))} {m['contig_num']}")
    return f"@{name} * {m['type']}" + " *" * 20


def py_pam_row(name, m, chrs):
    """write_pam_rec_pe, src/output.cpp:279-299 (note: 21 placeholders for an unmapped pair, 20 fields for a mapped one)."""
    if int(m["type"]) in MAPPED:
        cn = chrs[int(m["chr_id"])][0] if m["chr_id"] >= 0 else "-"
        f = [name, cn, m["spos_r1"], m["epos_r1"], m["mlen_r1"], m["qspos_r1"], m["qepos_r1"], "+" if m["r1_forward"] else "-", m["ed_r1"],
             cn, m["spos_r2"], m["epos_r2"], m["mlen_r2"], m["qspos_r2"], m["qepos_r2"], "+" if m["r2_forward"] else "-", m["ed_r2"],
             m["tlen"], m["junc_num"], int(bool(m["gm_compatible"])), m["type"]]
        return "\t".join(str(x) for x in f)
    return name + "\t*" * 21 + f"\t{m['type']}"


def _rand_states(rng, n, max_ed=4):
    st = np.zeros(n, dtype=cl.MAPPED_DTYPE)
    for i in range(n):
        t = int(rng.integers(0, 14))
        m = st[i]
        if t in MAPPED:
            m["type"] = t
            m["chr_id"] = int(rng.integers(0, len(CHRS)))
            for k in ("spos_r1", "spos_r2", "epos_r1", "epos_r2"):
                m[k] = int(rng.integers(1, 4000))
            for k in ("qspos_r1", "qspos_r2", "qepos_r1", "qepos_r2", "mlen_r1", "mlen_r2"):
                m[k] = int(rng.integers(1, 151))
            m["ed_r1"], m["ed_r2"] = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            m["tlen"] = int(rng.integers(-50, 5000))
            m["junc_num"] = int(rng.integers(0, 4))
            m["r1_forward"], m["r2_forward"], m["gm_compatible"] = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
            m["contig_num"] = CHRS[int(m["chr_id"])][1] - 1
        else:                                  # what finish_round / fill_map_info leave for an unmapped type
            m["type"] = t
            m["ed_r1"] = m["ed_r2"] = max_ed + 1
            m["tlen"] = 1000000000
            m["chr_id"] = -1
            m["r1_forward"] = m["r2_forward"] = 1
    return st


@pytest.mark.parametrize("gz", [False, True])
def test_parse_fresh_reads(built, tmp_path, gz):
    rng = np.random.default_rng(7)
    n = 2500
    s1, q1 = _rand_reads(rng, n)
    s2, q2 = _rand_reads(rng, n)
    names1 = [f"read{i}/1 extra comment" if i % 3 == 0 else (f"r{i}/1" if i % 3 == 1 else f"x{i}") for i in range(n)]
    names2 = [nm.replace("/1", "/2") for nm in names1]
    p1, p2 = str(tmp_path / ("a_1.fq" + (".gz" if gz else ""))), str(tmp_path / ("a_2.fq" + (".gz" if gz else "")))
    op = gzip.open if gz else open
    _fastq(p1, names1, s1, q1, op)
    _fastq(p2, names2, s2, q2, op)
    rd = cl.FastqReader(p1, p2, CHRS)
    got = 0
    while True:
        b = rd.next_batch(700)
        if b is None:
            break
        assert b.prior is None                                  # fresh reads: the default first-round state
        for i in range(b.n):
            g = got + i
            want = names1[g].split(" ")[0]
            want = want[:-2] if len(want) >= 2 and want[-2] == "/" else want
            assert b.name(i) == want and b.name(i, 2) == want
            assert b.seq(i).decode() == s1[g] and b.seq(i, 2).decode() == s2[g]
            assert b.qual(i).decode() == q1[g] and b.qual(i, 2).decode() == q2[g]
        got += b.n
    assert got == n
    rd.close()


def test_remain_and_pam_formats_round_trip(built, tmp_path):
    rng = np.random.default_rng(11)
    n = 1200
    s1, q1 = _rand_reads(rng, n)
    s2, q2 = _rand_reads(rng, n)
    names = [f"pair{i}" for i in range(n)]
    p1, p2 = str(tmp_path / "in_1.fq"), str(tmp_path / "in_2.fq")
    _fastq(p1, [x + "/1" for x in names], s1, q1)
    _fastq(p2, [x + "/2" for x in names], s2, q2)
    rd = cl.FastqReader(p1, p2, CHRS)
    b = rd.next_batch(n)
    st = _rand_states(rng, n)
    sel = np.sort(rng.choice(n, 800, replace=False)).astype(np.uint64)
    r1, r2, pam = str(tmp_path / "out_1_remain_R1.fastq"), str(tmp_path / "out_1_remain_R2.fastq"), str(tmp_path / "out.mapping.pam")
    w = cl.RecordWriter(r1, r2, CHRS)
    w.write_remain(b, st, sel)
    w.close()
    w = cl.RecordWriter(pam, None, CHRS)
    w.write_pam(b, st)
    w.close()
    # formats
    for path, seqs, quals in ((r1, s1, q1), (r2, s2, q2)):
        lines = open(path).read().split("\n")
        assert lines[-1] == "" and len(lines) == 4 * len(sel) + 1
        for k, i in enumerate(sel):
            i = int(i)
            assert lines[4 * k] == py_remain_header(names[i], st[i], CHRS)
            assert lines[4 * k + 1:4 * k + 4] == [seqs[i], "+", quals[i]]
            assert len(lines[4 * k].split(" ")) == 23
    rows = open(pam).read().split("\n")
    assert rows[-1] == "" and rows[:-1] == [py_pam_row(names[i], st[i], CHRS) for i in range(n)]
    # the carried state survives the file round trip bit for bit
    rd2 = cl.FastqReader(r1, r2, CHRS)
    b2 = rd2.next_batch(n)
    assert b2.n == len(sel) and b2.prior is not None
    assert b2.prior.tobytes() == st[sel.astype(np.int64)].tobytes()
    for k in (0, len(sel) // 2, len(sel) - 1):
        assert b2.name(k) == names[int(sel[k])] and b2.seq(k, 2).decode() == s2[int(sel[k])]
    rd.close()
    rd2.close()


def py_sam_rows(name, m, chrs, s1, q1, s2, q2):
    """set_flag_pe / set_output_pe / write_sam_rec_pe, src/output.cpp:118-277."""
    comp = dict(zip("ACGTNacgtn", "TGCANTGCAN"))
    t = int(m["type"])
    un = not (t <= 2 or t in (5, 7))
    rows = []
    for first in (True, False):
        flag = 1 | (2 if t == 0 else 0) | (12 if un else 0)
        mine_fw, mate_fw = (m["r1_forward"], m["r2_forward"]) if first else (m["r2_forward"], m["r1_forward"])
        if not un and not mine_fw:
            flag |= 16
        if not un and not mate_fw:
            flag |= 32
        flag |= 64 if first else 128
        tl = int(m["tlen"]) if (m["spos_r1"] < m["spos_r2"]) == first else -int(m["tlen"])
        seq, qual = (s1, q1) if first else (s2, q2)
        if flag & 16:
            seq, qual = "".join(comp[c] for c in reversed(seq)), qual[::-1]
        if un:
            f = [name, flag, "*", 0, 255, "*", "*", 0, 0, seq, qual, f"AT:i:{t}", "NM:i:0", "JC:i:0", "TC:i:0"]
        else:
            cn = chrs[int(m["chr_id"])][0]
            pos, pnext = (m["spos_r1"], m["spos_r2"]) if first else (m["spos_r2"], m["spos_r1"])
            ed = m["ed_r1"] if first else m["ed_r2"]
            f = [name, flag, cn, pos, 255, "*", "=", pnext, tl & 0xFFFFFFFF, seq, qual, f"AT:i:{t}", f"NM:i:{ed}", f"JC:i:{m['junc_num']}",
                 f"TC:i:{int(bool(m['gm_compatible']))}"]
        rows.append("\t".join(str(x) for x in f))
    return rows


def test_sam_header_and_records(built, tmp_path):
    rng = np.random.default_rng(23)
    n = 900
    s1, q1 = _rand_reads(rng, n)
    s2, q2 = _rand_reads(rng, n)
    names = [f"frag{i}" for i in range(n)]
    p1, p2 = str(tmp_path / "in_1.fq"), str(tmp_path / "in_2.fq")
    _fastq(p1, names, s1, q1)
    _fastq(p2, names, s2, q2)
    rd = cl.FastqReader(p1, p2, CHRS)
    b = rd.next_batch(n)
    st = _rand_states(rng, n)
    sam = str(tmp_path / "out.mapping.sam")
    w = cl.RecordWriter(sam, None, CHRS)
    w.write_sam_header()
    w.write_sam(b, st)
    sel = np.array([5, 17, 899], dtype=np.uint64)
    w.write_sam(b, st, sel)
    w.close()
    rows = open(sam).read().split("\n")
    assert rows[-1] == ""
    assert rows[:4] == ["@HD\tVN:1.4\tSO:unsorted"] + [f"@SQ\tSN:{c[0]}\tLN:{c[3]}" for c in CHRS]
    want = []
    for i in list(range(n)) + [5, 17, 899]:
        want += py_sam_rows(names[i], st[i], CHRS, s1[i], q1[i], s2[i], q2[i])
    assert rows[4:-1] == want
    kinds = {int(r.split("\t")[1]) for r in rows[4:-1]}
    assert {77, 141} <= kinds and any(k & 16 for k in kinds) and any(k & 2 for k in kinds)
    rd.close()


def test_malformed_fastq_is_an_error_not_a_crash(built, tmp_path):
    p1, p2 = str(tmp_path / "bad_1.fq"), str(tmp_path / "bad_2.fq")
    open(p1, "w").write("@a\nACGT\n+\nIIII\n@b\nACGT\n-\nIIII\n")
    open(p2, "w").write("@a\nACGT\n+\nIIII\n@b\nACGT\n+\nIII\n")
    rd = cl.FastqReader(p1, p2)
    with pytest.raises(RuntimeError):
        rd.next_batch(10)
    rd.close()
    with pytest.raises(RuntimeError):
        cl.FastqReader(str(tmp_path / "missing_1.fq"), p2)
    # R2 shorter than R1 is an error; R2 longer is read up to R1's end (the reference stops when R1 runs out)
    rec = lambda names: "".join(f"@{x}\nACGTA\n+\nIIIII\n" for x in names)
    open(p1, "w").write(rec("abc"))
    open(p2, "w").write(rec("ab"))
    rd = cl.FastqReader(p1, p2)
    with pytest.raises(RuntimeError):
        rd.next_batch(10)
    rd.close()
    open(p2, "w").write(rec("abcde"))
    rd = cl.FastqReader(p1, p2)
    b = rd.next_batch(10)
    assert b.n == 3 and [b.name(i, 2) for i in range(3)] == list("abc") and b.seq(2, 2) == b"ACGTA" and b.qual(2, 2) == b"IIIII"
    rd.close()


def test_parallel_plain_text_path_equals_the_serial_parser(built, tmp_path, monkeypatch):
    """Plain-text input is read in blocks and tokenised on several threads (newline index -> records -> prefix sums -> parallel
    copies); gzip input and CM_FASTQ_SERIAL=1 take the record-by-record parser.  Same batches either way: ragged reads, carried
    23-token headers mixed with fresh ones, '/1' suffixes, extra header tokens, batch boundaries that cut the block anywhere, a
    last record without a newline, R2 longer than R1."""
    rng = np.random.default_rng(12)
    n = 5000
    seqs1, quals1 = _rand_reads(rng, n, 20, 301)
    seqs2, quals2 = _rand_reads(rng, n + 7, 20, 301)
    st = _rand_states(rng, n)
    names1, names2 = [], []
    for i in range(n):
        base = f"read{i}" if i % 3 else f"r{i}/1"
        if i % 5 == 0:
            names1.append(py_remain_header(f"read{i}", st[i], CHRS)[1:])
        else:
            names1.append(base + (" extra tokens here" if i % 7 == 0 else ""))
    for i in range(n + 7):
        names2.append(f"read{i}/2")
    p1, p2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    _fastq(p1, names1, seqs1, quals1)
    _fastq(p2, names2, seqs2, quals2)
    with open(p1, "rb+") as f:                 # drop the final newline of R1
        f.seek(-1, 2)
        f.truncate()

    def read_all(batch):
        rd = cl.FastqReader(p1, p2, CHRS, 4)
        out = []
        while True:
            b = rd.next_batch(batch)
            if b is None:
                break
            out.append((b.n, [b.name(i) for i in range(b.n)], [b.name(i, 2) for i in range(b.n)], [b.seq(i) for i in range(b.n)], [b.seq(i, 2) for i in range(b.n)],
                        [b.qual(i) for i in range(b.n)], [b.qual(i, 2) for i in range(b.n)], None if b.prior is None else b.prior.tobytes()))
        rd.close()
        return out

    for batch in (n + 100, 777, 1):
        if batch == 1:
            continue
        monkeypatch.delenv("CM_FASTQ_SERIAL", raising=False)
        monkeypatch.setenv("CM_FASTQ_THREADS", "6")
        fast = read_all(batch)
        monkeypatch.setenv("CM_FASTQ_SERIAL", "1")
        slow = read_all(batch)
        assert sum(x[0] for x in fast) == n and len(fast) == len(slow)
        assert fast == slow
    assert fast[0][1][0] == "read0" and fast[0][3][1] == seqs1[1].encode()
    # malformed input is refused by both
    bad = str(tmp_path / "bad_1.fq")
    open(bad, "w").write("@x\nACGT\n+\nIII\n")              # quality shorter than the sequence
    for serial in (None, "1"):
        if serial:
            monkeypatch.setenv("CM_FASTQ_SERIAL", serial)
        else:
            monkeypatch.delenv("CM_FASTQ_SERIAL", raising=False)
        rd = cl.FastqReader(bad, bad, CHRS, 4)
        with pytest.raises(RuntimeError):
            rd.next_batch(10)
        rd.close()


def test_pam_rows_formatted_in_parallel_are_the_same_bytes(built, tmp_path, monkeypatch):
    """batches of >= 65536 rows are formatted on several threads (private buffers, written in order)"""
    rng = np.random.default_rng(3)
    n = 70000
    seqs, quals = ["ACGT"] * n, ["IIII"] * n
    names = [f"q{i}" for i in range(n)]
    p1, p2 = str(tmp_path / "b_1.fq"), str(tmp_path / "b_2.fq")
    _fastq(p1, names, seqs, quals)
    _fastq(p2, names, seqs, quals)
    st = _rand_states(rng, n)
    rd = cl.FastqReader(p1, p2, CHRS, 4)
    b = rd.next_batch(n)
    out = []
    for threads in ("1", "5"):
        monkeypatch.setenv("CM_WRITER_THREADS", threads)
        path = str(tmp_path / f"o{threads}.pam")
        w = cl.RecordWriter(path, None, CHRS)
        w.write_pam(b, st)
        w.write_pam(b, st, np.arange(0, n, 3))
        w.close()
        out.append(open(path, "rb").read())
    rd.close()
    assert out[0] == out[1] and out[0].count(b"\n") == n + len(range(0, n, 3))
    assert out[0].split(b"\n")[5].decode() == py_pam_row("q5", st[5], CHRS)


def test_write_errors_are_reported(built, tmp_path):
    """a full device must not leave a silently truncated file behind (/dev/full accepts the open and fails every write)"""
    import os
    if not os.path.exists("/dev/full"):
        pytest.skip("no /dev/full")
    rng = np.random.default_rng(4)
    n = 3000
    names = [f"w{i}" for i in range(n)]
    p1, p2 = str(tmp_path / "c_1.fq"), str(tmp_path / "c_2.fq")
    _fastq(p1, names, ["ACGTACGT"] * n, ["IIIIIIII"] * n)
    _fastq(p2, names, ["ACGTACGT"] * n, ["IIIIIIII"] * n)
    rd = cl.FastqReader(p1, p2, CHRS, 4)
    b = rd.next_batch(n)
    st = _rand_states(rng, n)
    w = cl.RecordWriter("/dev/full", None, CHRS)
    w.write_pam(b, st)                      # buffered: may not notice yet
    with pytest.raises(RuntimeError):
        w.close()
    w = cl.RecordWriter(str(tmp_path / "ok.pam"), None, CHRS)
    w.write_pam(b, st)
    w.close()
    rd.close()


def _call_log(r1, r2, batch, serial, monkeypatch):
    """(pairs or 'error') of every cm_fastq_next call until the end of input or the first error"""
    if serial:
        monkeypatch.setenv("CM_FASTQ_SERIAL", "1")
    else:
        monkeypatch.delenv("CM_FASTQ_SERIAL", raising=False)
    rd = cl.FastqReader(r1, r2, CHRS, 4)
    log = []
    while True:
        try:
            b = rd.next_batch(batch)
        except RuntimeError:
            log.append("error")
            break
        if b is None:
            break
        log.append(b.n)
    rd.close()
    return log


@pytest.mark.parametrize("tail", ["\n", "\n\n\n", "\n\n\n\n", "@partial\nACGT\n", "@partial\nACGT\n+\n"])
def test_trailing_garbage_is_refused_alike_by_both_parsers(built, tmp_path, monkeypatch, tail):
    """What follows the last whole record (blank lines, a partial record) is malformed: the reference asserts the '@' of the next
    record (src/fastq_parser.h:64-65).  The chunk-parallel plain-text path and the record-by-record path (gzip / pipes) give
    the same verdict in the same cm_fastq_next call, for any batch size, on either mate's file."""
    rng = np.random.default_rng(5)
    n = 40
    seqs, quals = _rand_reads(rng, n, 30, 151)
    names = [f"r{i}" for i in range(n)]
    good, dirty = str(tmp_path / "good.fq"), str(tmp_path / "dirty.fq")
    _fastq(good, names, seqs, quals)
    _fastq(dirty, names, seqs, quals)
    with open(dirty, "a") as f:
        f.write(tail)
    for r1, r2 in ((dirty, good), (good, dirty)):
        for batch in (n + 5, n, 16, 7):
            fast = _call_log(r1, r2, batch, False, monkeypatch)
            slow = _call_log(r1, r2, batch, True, monkeypatch)
            assert fast == slow, (r1 == dirty, batch, fast, slow)
            assert fast[-1] == "error"


def test_fifo_and_gzip_fifo_input(built, tmp_path):
    """Non-seekable input (named pipe, process substitution, /dev/stdin) is read through zlib on the open descriptor, plain text
    and gzip alike, as the reference's gzopen / gzread does for every input (src/fastq_parser.cpp:60,85) -- no magic sniff,
    no pread(), no reopening."""
    import gzip
    import threading
    rng = np.random.default_rng(8)
    n = 3000
    seqs, quals = _rand_reads(rng, n, 30, 151)
    names = [f"r{i}" for i in range(n)]
    plain = str(tmp_path / "p.fq")
    _fastq(plain, names, seqs, quals)
    data = open(plain, "rb").read()
    for payload in (data, gzip.compress(data, 1)):
        f1, f2 = str(tmp_path / "f1"), str(tmp_path / "f2")
        for f in (f1, f2):
            if os.path.exists(f):
                os.unlink(f)
            os.mkfifo(f)

        def feed(path):
            with open(path, "wb") as w:
                w.write(payload)

        th = [threading.Thread(target=feed, args=(f,)) for f in (f1, f2)]
        [t.start() for t in th]
        rd = cl.FastqReader(f1, f2, CHRS, 4)
        got = 0
        while True:
            b = rd.next_batch(1024)
            if b is None:
                break
            assert [b.seq(i) for i in (0, b.n - 1)] == [seqs[got].encode(), seqs[got + b.n - 1].encode()]
            assert b.name(b.n - 1, 2) == names[got + b.n - 1]
            got += b.n
        rd.close()
        [t.join() for t in th]
        assert got == n


def test_sharded_reader_blocks(built, tmp_path):
    """cm_fastq_open_shard: rank r of W gets the pairs [r*N/W, (r+1)*N/W) of both files, cut at the same record
    although the two files have different record sizes; the blocks of all ranks, in rank order, are the unsharded input.  Ragged
    reads, last record without a newline, R2 longer than R1, more ranks than pairs; gzip input gives the same blocks."""
    import gzip
    rng = np.random.default_rng(3)
    n = 4001
    seqs1, quals1 = _rand_reads(rng, n, 20, 301)
    seqs2, quals2 = _rand_reads(rng, n + 3, 20, 301)
    p1, p2 = str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")
    _fastq(p1, [f"frag{i} x" for i in range(n)], seqs1, quals1)
    _fastq(p2, [f"frag{i}/2" for i in range(n + 3)], seqs2, quals2)
    with open(p1, "rb+") as f:
        f.seek(-1, 2)
        f.truncate()

    def block(rank, world):
        rd = cl.FastqReader(p1, p2, CHRS, 4, rank=rank, world=world, n_threads=3)
        out = []
        while True:
            b = rd.next_batch(700)
            if b is None:
                break
            out += [(b.name(i), b.seq(i), b.qual(i), b.name(i, 2), b.seq(i, 2)) for i in range(b.n)]
        first, cnt = rd.first_pair, rd.n_pairs
        rd.close()
        return out, first, cnt

    whole, _, _ = block(0, 1)
    assert len(whole) == n and whole[-1][1] == seqs1[-1].encode()
    for world in (2, 3, 7):
        got, at = [], 0
        for r in range(world):
            part, first, cnt = block(r, world)
            assert first == at == n * r // world and cnt == len(part) == n * (r + 1) // world - n * r // world
            got += part
            at += cnt
        assert got == whole
    tiny1, tiny2 = str(tmp_path / "t_1.fq"), str(tmp_path / "t_2.fq")
    _fastq(tiny1, ["a", "b"], seqs1[:2], quals1[:2])
    _fastq(tiny2, ["a", "b"], seqs2[:2], quals2[:2])
    sizes = []
    for r in range(5):
        rd = cl.FastqReader(tiny1, tiny2, CHRS, 4, rank=r, world=5)
        b = rd.next_batch(10)
        sizes.append(0 if b is None else b.n)
        rd.close()
    assert sum(sizes) == 2 and max(sizes) == 1
    with pytest.raises(RuntimeError):
        cl.FastqReader(tiny1, tiny2, CHRS, 4, rank=2, world=2)
    # gzip input (the reference reads .gz everywhere): the same blocks, found by counting R1's records with an inflate pass; a mix
    # of one gzip and one plain file works too (both then go through the record-by-record parser)
    g1, g2 = str(tmp_path / "s_1.fq.gz"), str(tmp_path / "s_2.fq.gz")
    for src, dst in ((p1, g1), (p2, g2)):
        with gzip.open(dst, "wb") as f:
            f.write(open(src, "rb").read())
    p1_plain, p2_plain = p1, p2
    for a, b in ((g1, g2), (g1, p2_plain)):
        p1, p2 = a, b
        for world in (1, 3):
            got, at = [], 0
            for r in range(world):
                part, first, cnt = block(r, world)
                if world > 1:
                    assert first == at == n * r // world and cnt == len(part) == n * (r + 1) // world - n * r // world
                got += part
                at += len(part)
            assert got == whole, (a, b, world)
    p1, p2 = p1_plain, p2_plain


@pytest.mark.parametrize("serial", [False, True])
def test_batch_arrays_that_grow_are_announced_before_they_go(tmp_path, monkeypatch, serial):
    """A caller page-locks the arrays of a batch (cm_mapping_run does, with cm_host_register).  A generation's array that has to
    grow -- reads that get longer from batch to batch -- is allocated anew (never moved in place), and the release hook names the
    old block before it is freed: every array of a generation is either the block handed out for that generation's previous batch
    or a new one whose predecessor has been announced; closing announces the rest."""
    import ctypes as C
    if serial:
        monkeypatch.setenv("CM_FASTQ_SERIAL", "1")
    rng = np.random.default_rng(7)
    per, nb = 300, 12
    names, s1, q1, s2, q2 = [], [], [], [], []
    for b in range(nb):                                      # batch b: reads of (40 + 60 b) bases: capacities are passed several times
        for i in range(per):
            L = 40 + 60 * b
            names.append(f"r{b}_{i}")
            s1.append("".join(rng.choice(list("ACGT"), L)))
            s2.append("".join(rng.choice(list("ACGT"), L + 3)))
            q1.append("I" * L)
            q2.append("I" * (L + 3))
    p1, p2 = str(tmp_path / "g1.fq"), str(tmp_path / "g2.fq")
    _fastq(p1, names, s1, q1)
    _fastq(p2, names, s2, q2)
    rd = cl.FastqReader(p1, p2)
    live = set()                                             # blocks handed out and not announced as going
    gone = []

    def hook(user, ptr, nbytes):
        live.discard(ptr)
        gone.append((ptr, nbytes))

    cb = cl.RELEASE_HOOK(hook)
    rd.L.cm_fastq_set_release_hook(rd.h, cb, None)
    seen = 0
    per_gen = {}
    for b in range(nb):
        pb = rd.next_batch(per)
        assert pb.n == per
        mine = [C.cast(ptr, C.c_void_p).value for ptr in (pb.c.seq1, pb.c.seq2, pb.c.off1, pb.c.off2)]
        if b >= 4:                                           # this generation's previous blocks: kept, or announced before the new ones came
            for old, new in zip(per_gen[b % 4], mine):
                assert old == new or old not in live
        per_gen[b % 4] = mine
        live.update(mine)
        assert pb.seq(0).decode() == s1[b * per] and pb.seq(per - 1, 2).decode() == s2[b * per + per - 1]
        seen += pb.n
    assert rd.next_batch(per) is None or rd.next_batch(per).n == 0
    assert len(gone) >= 8                                    # four generations x two sequence arrays outgrew their first blocks
    n_before = len(gone)
    rd.close()
    assert len(gone) > n_before and not live                 # closing announces what is left
