"""SURVEY §8(f) N3, first and last step of stage 2 (circminer_amd/csrc/host_circ.cpp): the remain-FASTQ sort against GNU
sort itself (the reference shells out to it, src/process_circ.cpp:179-193) and report_events against a Python
restatement (src/process_circ.cpp:1554-1631, src/common.cpp:479-493, src/utils.cpp:771-816).  The BSJ calling between
the two is not built."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from circminer_amd import lib as cl


def _remain_file(path, rng, n):
    with open(path, "w") as f:
        for i in range(n):
            kind = rng.integers(0, 10)
            gs = int(rng.choice([5, 17, 100, 1000000, 1100000123, 2200004567, 3299999999])) if kind < 7 else int(rng.integers(0, 4_000_000_000))
            tok = str(gs) if kind != 9 else "*"                       # unmapped header: non-numeric key = 0
            name = f"r{int(rng.integers(0, n // 3 + 1))}.{i}" if kind % 2 else f"R{i}"
            L = int(rng.integers(20, 60))
            seq = "".join(rng.choice(list("ACGTN"), L))
            qual = "".join(chr(int(x)) for x in rng.integers(33, 74, L))
            f.write(f"@{name} {tok} 3 chr1 10 20 30 1 30 + 0 chr1 40 50 30 1 30 - 1 200 0 1 {gs // 1100000000}\n{seq}\n+\n{qual}\n")


@pytest.mark.skipif(not all(shutil.which(x) for x in ("paste", "sort", "tr")), reason="GNU coreutils not on PATH")
def test_sort_remain_equals_gnu_sort_pipeline(built, tmp_path):
    rng = np.random.default_rng(4)
    p = str(tmp_path / "o_3_remain_R1.fastq")
    _remain_file(p, rng, 4000)
    got = cl.sort_remain(p)
    want = p + ".gnu"
    env = dict(os.environ, LC_ALL="C")
    subprocess.check_call(f'cat {p} | paste - - - - | sort -S 64M -k2,2n | tr "\\t" "\\n" > {want}', shell=True, env=env)
    assert open(got, "rb").read() == open(want, "rb").read()
    keys = [ln.split(" ")[1] for ln in open(got).read().split("\n")[0::4] if ln]
    nums = [int(k) if k != "*" else 0 for k in keys]
    assert nums == sorted(nums) and "*" in keys and len(nums) == 4000
    empty = str(tmp_path / "empty.fastq")
    open(empty, "w").close()
    assert open(cl.sort_remain(empty)).read() == ""


def test_sort_remain_corner_cases_equal_gnu_sort(built, tmp_path):
    """What `paste - - - - | sort -k2,2n | tr` does to input the mapping stage never writes but a user file may hold: ties on the
    key (whole-line byte order decides, line ends compared as the TABs paste makes of them), negative / zero-padded / absent /
    non-numeric keys, a header of one field (field 2 then starts with the sequence line), an incomplete last group, a last line
    without a newline."""
    recs = ["@a 5 x\nACGT\n+\nIIII", "@b 5 x\nACGT\n+\nIIII", "@a 5 w\nACGT\n+\nIIII", "@c -3 y\nAC\n+\nII", "@d 0007 y\nAC\n+\nII",
            "@e * z\nAC\n+\nII", "@f\nAC\n+\nII", "@g 12abc q\nAC\n+\nII", "@h -0 q\nAC\n+\nII", "@i 7 q\nAC\n+\nII", "@a 5 x\nACGT\n+\nIIIJ",
            "@j  9 two blanks\nAC\n+\nII", "@k 99999999999999999999 big\nAC\n+\nII", "@l 100000000000000000000 bigger\nAC\n+\nII"]
    env = dict(os.environ, LC_ALL="C")
    for tail in ("\n", "", "\n@z 1 partial\nAC\n", "\n@z 1 partial\nAC"):
        p = str(tmp_path / f"corner{abs(hash(tail))}.fastq")
        with open(p, "w") as f:
            f.write("\n".join(recs) + tail)
        got = cl.sort_remain(p)
        want = p + ".gnu"
        subprocess.check_call(f'cat {p} | paste - - - - | sort -S 64M -k2,2n | tr "\\t" "\\n" > {want}', shell=True, env=env)
        assert open(got, "rb").read() == open(want, "rb").read(), tail


def py_consensus(seqs):
    if not seqs or any(len(s) != len(seqs[0]) for s in seqs):
        return ""
    out = ""
    for i in range(len(seqs[0])):
        cnt = {b: sum(s[i].upper() == b for s in seqs) for b in "ACGT"}
        best, ch = 0, "N"
        for b in "ACGT":
            if cnt[b] > best:
                best, ch = cnt[b], b
        out += ch if best >= len(seqs) // 2 else "N"
    return out


def test_circ_report_rows(built, tmp_path):
    rng = np.random.default_rng(8)
    events = [("chr1", 1000, 5000, "AG", "GT"), ("chr1", 1000, 6000, "AG", "GT"), ("chr10", 50, 900, "AC", "GT"), ("chr2", 7, 70, "AG", "GC"),
              ("chr2", 7, 71, "AG", "GT"), ("chrX", 123456, 223456, "AG", "GT")]
    calls = []
    for k in range(300):
        e = events[int(rng.integers(0, len(events)))]
        t = int(rng.choice([20, 20, 20, 21, 22]))
        if e[:3] == ("chr2", 7, 71):
            t = 21                                                    # an event seen only as NCR: never printed
        sig = lambda ref: "".join(c if rng.random() < 0.8 else str(rng.choice(list("ACGTacgtN"))) for c in ref)
        ss, es = sig(e[3]), sig(e[4])
        if e[:3] == ("chr10", 50, 900) and rng.random() < 0.3:
            ss = ss + "A"                                             # ragged signals: consensus is empty -> Fail
        calls.append((e[0], f"read{k}", e[1], e[2], t, ss, es, e[3], e[4]))
    path = str(tmp_path / "out.circ_report")
    cl.circ_report(calls, path)
    rows = [r.split("\t") for r in open(path).read().strip().split("\n")]
    groups = {}
    for c in calls:
        groups.setdefault((c[0], c[2], c[3]), []).append(c)
    want_keys = sorted(k for k, g in groups.items() if min(x[4] for x in g) == 20)
    assert [(r[0], int(r[1]), int(r[2])) for r in rows] == want_keys and ("chr2", 7, 71) not in want_keys
    for r in rows:
        g = groups[(r[0], int(r[1]), int(r[2]))]
        ss, es = py_consensus([x[5] for x in g]), py_consensus([x[6] for x in g])
        assert int(r[3]) == len(g) and r[4] == "STC" and r[5] == f"{ss}-{es}" and r[6] == f"{g[0][7]}-{g[0][8]}"
        assert r[7] == ("Pass" if (ss, es) == (g[0][7], g[0][8]) else "Fail")
        assert sorted(r[8].split(",")) == sorted(x[1] for x in g)     # order inside the row is std::sort's (unstable)
    assert {r[7] for r in rows} == {"Pass", "Fail"}
    cl.circ_report([], path)
    assert open(path).read() == ""
    # a full device is an error (CM_EIO), not a silently truncated report; the sorted remain file likewise
    import os
    if os.path.exists("/dev/full"):
        with pytest.raises(RuntimeError):
            cl.circ_report(calls, "/dev/full")
        fq = str(tmp_path / "x.fastq")
        open(fq, "w").write("".join(f"@r{i} {1000 - i} 3 chr1\nACGT\n+\nIIII\n" for i in range(50000)))
        with pytest.raises(RuntimeError):
            cl.sort_remain(fq, "/dev/full")


def test_regional_hash_table(built):
    """RegionalHashTable::create_table (src/hash_table.cpp:58-112) flattened to CSR: every 8-mer of a gene region with its
    location, lower case accepted, windows with N skipped, buckets above MAXHIT = 1000 emptied."""
    import ctypes as C
    L = cl.load()
    rng = np.random.default_rng(6)
    ws = 8
    seq = "".join(rng.choice(list("ACGT"), 30000)) + "ACGTACGT" * 300 + "acgtnNacGT" * 50 + "A" * 1500 + "".join(rng.choice(list("ACGTacgt"), 2000))
    b = np.frombuffer(seq.encode(), np.uint8)
    off, loc = cl.u32p(), cl.u32p()
    start = 123456
    assert L.cm_regional_table_build(b.ctypes.data, start, len(b), ws, C.byref(off), C.byref(loc)) == 0
    offs = np.ctypeslib.as_array(off, shape=(4 ** ws + 1,)).copy()
    locs = np.ctypeslib.as_array(loc, shape=(max(int(offs[-1]), 1),))[:int(offs[-1])].copy()
    L.cm_regional_table_free(off, loc)
    want = {}
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    up = seq.upper()
    for i in range(len(seq) - ws + 1):
        w = up[i:i + ws]
        if any(c not in code for c in w):
            continue
        hv = 0
        for c in w:
            hv = hv * 4 + code[c]
        want.setdefault(hv, []).append(start + i)
    dropped = [hv for hv, v in want.items() if len(v) > 1000]
    assert dropped and 0 in dropped                                 # the poly-A run; ACGTACGT x 300 stays (<= 1000 each)
    for hv in range(4 ** ws):
        got = list(locs[offs[hv]:offs[hv + 1]])
        exp = want.get(hv, [])
        assert got == ([] if len(exp) > 1000 else exp), hv
    # shorter than a window / empty
    assert L.cm_regional_table_build(b.ctypes.data, 0, ws - 1, ws, C.byref(off), C.byref(loc)) == 0
    assert np.ctypeslib.as_array(off, shape=(4 ** ws + 1,))[-1] == 0
    L.cm_regional_table_free(off, loc)
