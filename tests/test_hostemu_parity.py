"""The kernel bodies of circminer_amd/csrc/cm_core.h, run lane-by-lane on the CPU by
tests/hostemu.cpp, against the oracle (bit-exact).  This is how the device code is debugged on the
build box, which has no GPU; the same comparisons run on the real HIP path in test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

from circminer_amd import lib as cl, synth
from oracle import oracle_py as op
from conftest import first_diff


def _emu_rounds(E, ds, P):
    st0, act0 = op.default_state(P, ds.batch.n)
    st1, act1 = st0.copy(), act0.copy()
    for ci in range(ds.hi.n_contigs):
        last = ci == ds.hi.n_contigs - 1
        iv, av = ds.hi.views[ci], ds.hi.annots[ci]                          # the product's builders feed the kernel bodies,
        cat0 = op.map_round(P, ds.ohi.views[ci], ds.ohi.annots[ci], ds.batch, last, st0, act0)      # the oracle's own feed the oracle
        cat1 = np.full(ds.batch.n, -1, np.int32)
        rc = E.emu_map_round(C.byref(P), C.byref(iv), C.byref(av), C.byref(ds.batch.c), int(last), st1.ctypes.data, act1.ctypes.data,
                             cat1.ctypes.data)
        assert rc == 0
        assert (cat0 == cat1).all() and (act0 == act1).all()
        assert st0.tobytes() == st1.tobytes(), first_diff(st0, st1)


def test_seeds_and_chains(emu, ds_tiny):
    P = cl.default_params()
    iv, av, b = ds_tiny.hi.views[0], ds_tiny.hi.annots[0], ds_tiny.batch
    S = b.max_len() // P.kmer
    a0, b0, c0 = op.seeds(P, iv, b, S)
    a1, b1, c1 = (np.zeros_like(a0) for _ in range(3))
    assert emu.emu_seed_batch(C.byref(P), C.byref(iv), C.byref(b.c), S, a1.ctypes.data, b1.ctypes.data, c1.ctypes.data) == 0
    assert (c0 == c1).all() and (b0 == b1).all() and (a0[c0 > 0] == a1[c0 > 0]).all()
    ch0, n0, h0 = op.chains(P, iv, av, b)
    ch1, n1, h1 = np.zeros_like(ch0), np.zeros_like(n0), np.zeros_like(h0)
    assert emu.emu_chain_batch(C.byref(P), C.byref(iv), C.byref(av), C.byref(b.c), ch1.ctypes.data, n1.ctypes.data, h1.ctypes.data) == 0
    assert (n0 == n1).all() and (h0 == h1).all()
    x, y = ch0.reshape(-1, 30), ch1.reshape(-1, 30)
    for r in np.nonzero(n0)[0]:
        for k in range(n0[r]):
            L = int(x[r, k]["chain_len"])
            assert L == int(y[r, k]["chain_len"]) and x[r, k]["score"] == y[r, k]["score"]
            assert (x[r, k]["rpos"][:L] == y[r, k]["rpos"][:L]).all() and (x[r, k]["qpos"][:L] == y[r, k]["qpos"][:L]).all()


@pytest.mark.parametrize("name", ["ds_tiny", "ds_tiny2r"])
def test_map_rounds(emu, name, request):
    _emu_rounds(emu, request.getfixturevalue(name), cl.default_params())


@pytest.mark.parametrize("kw", [dict(scan_level=1), dict(scan_level=2, max_ed=8, seed_lim=1000), dict(band=2), dict(band=5, max_ed=6),
                                dict(max_chain_len=4, max_intron=5000), dict(max_sc=3, max_tlen=250)])
def test_map_rounds_param_variants(emu, ds_tiny2r, kw):
    _emu_rounds(emu, ds_tiny2r, cl.default_params(**kw))


def test_ragged_and_dirty_reads(emu, ds_dirty):
    """empty / sub-seed / ragged / 300-bp reads, N runs and lower-case stretches (see conftest.ds_dirty)"""
    _emu_rounds(emu, ds_dirty, cl.default_params())
    _emu_rounds(emu, ds_dirty, cl.default_params(scan_level=2, max_ed=6))


def test_k22_int16_checksum_quirk(emu, tmp_path_factory):
    """k=22: the reference compares an int16 target with the uint16 checksum (match_read.cpp:77), so
    k-mers whose checksum >= 0x8000 are never found.  Oracle and device code both reproduce it."""
    from conftest import DataSet
    ds = DataSet(tmp_path_factory.mktemp("k22"), "tiny", 600, 29, kmer=22)
    P = cl.default_params(kmer=22)
    iv, b = ds.hi.views[0], ds.batch
    S = b.max_len() // 22
    a0, b0, c0 = op.seeds(P, iv, b, S)
    a1, b1, c1 = (np.zeros_like(a0) for _ in range(3))
    assert emu.emu_seed_batch(C.byref(P), C.byref(iv), C.byref(b.c), S, a1.ctypes.data, b1.ctypes.data, c1.ctypes.data) == 0
    assert (c0 == c1).all()
    # forward-orientation probes of R1: found iff the 15th base of the seed is A or C
    r1 = ds.d.seq1
    probe = c0.reshape(b.n, 2, 2, S)[:, 0, 0, :]
    base15 = r1[:, [s * 22 + 14 for s in range(S)]]
    hi_half = np.isin(base15, [ord("G"), ord("T")])
    assert (probe[hi_half] == 0).all() and (probe[~hi_half] > 0).mean() > 0.3   # ~half the R1s are reverse-strand
    _emu_rounds(emu, ds, P)


def test_dp_bodies_fuzz(emu):
    O = op.load()
    rng = np.random.default_rng(11)
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    for band, max_ed in ((3, 4), (2, 4), (5, 8)):
        P = cl.default_params(band=band, max_ed=max_ed)
        for _ in range(700):
            m = int(rng.integers(1, 150))
            t = A[rng.integers(0, 4, m)].copy()
            s = t.copy()
            for _k in range(int(rng.integers(0, 5))):
                s[int(rng.integers(0, len(s)))] = A[rng.integers(0, 4)]
            if rng.random() < 0.4 and len(s) > 3:
                s = np.delete(s, int(rng.integers(0, len(s))))
            if rng.random() < 0.4:
                s = np.insert(s, int(rng.integers(0, len(s))), A[rng.integers(0, 4)])
            s = np.concatenate([s, A[rng.integers(0, 4, band + 2)]])[:m + band]
            if rng.random() < 0.15:
                s[int(rng.integers(0, len(s)))] = ord("N")
            if rng.random() < 0.15:
                t[int(rng.integers(0, m))] = ord("n")
            s, n = np.ascontiguousarray(s), len(s)
            for left in (0, 1):
                a = [C.c_int() for _ in range(3)]
                b = [C.c_int() for _ in range(3)]
                r0 = O.oracle_drop_sc(C.byref(P), s.ctypes.data, n, t.ctypes.data, m, left, *[C.byref(x) for x in a])
                r1 = emu.emu_drop_sc(C.byref(P), s.ctypes.data, n, t.ctypes.data, m, left, *[C.byref(x) for x in b])
                assert r0 == r1 and [x.value for x in a] == [x.value for x in b], (band, left, n, m)
                if n > m:
                    a, b = [C.c_int(), C.c_int()], [C.c_int(), C.c_int()]
                    r0 = O.oracle_edit_side(C.byref(P), s.ctypes.data, n, t.ctypes.data, m, left, C.byref(a[0]), C.byref(a[1]))
                    r1 = emu.emu_edit_side(C.byref(P), s.ctypes.data, n, t.ctypes.data, m, left, C.byref(b[0]), C.byref(b[1]))
                    assert r0 == r1 and [x.value for x in a] == [x.value for x in b], (band, left, n, m)
            w = int(rng.integers(0, band + 1))
            n3 = int(rng.integers(0, 60))
            s3 = A[rng.integers(0, 4, max(n3, 1))]
            t3 = np.ascontiguousarray(np.concatenate([s3[:n3], A[rng.integers(0, 4, w + 1)]])[:n3 + w + 1])
            assert O.oracle_one_side(s3.ctypes.data, n3, t3.ctypes.data, n3 + w, w) == emu.emu_one_side(C.byref(P), s3.ctypes.data, n3, t3.ctypes.data, n3 + w, w)


def test_xdrop_long_prefixes_and_low_complexity(emu):
    """X-drop DP on strings that share a long exact prefix, first difference anywhere (substitution, insertion, deletion,
    N), low-complexity alphabets (ties between the diagonal and shifted alignments), every length up to 300.
    (Written for a fast-forward over exact prefixes that was bit-exact but slower on MI355X; kept as a fuzz.)"""
    O = op.load()
    rng = np.random.default_rng(5)
    P = cl.default_params()
    for it in range(6000):
        alpha = [b"ACGT", b"AC", b"A", b"AAC", b"ACGTACGA"][it % 5]
        A = np.frombuffer(alpha, dtype=np.uint8)
        m = int(rng.integers(20, 300))
        t = A[rng.integers(0, len(A), m)].copy()
        s = t.copy()
        kind = it % 7
        p = int(rng.integers(0, m))
        if kind == 0:
            s[p] = ord("ACGT"[int(rng.integers(0, 4))])
        elif kind == 1:
            s = np.delete(s, p)
        elif kind == 2:
            s = np.insert(s, p, ord("ACGT"[int(rng.integers(0, 4))]))
        elif kind == 3:
            s[p] = ord("N")
        elif kind == 4:
            t[p] = ord("n")
        elif kind == 5:
            for q in rng.integers(p, m, 3):
                s[int(q)] = ord("ACGT"[int(rng.integers(0, 4))])
        # kind 6: identical (the wrapper's closed form) plus the tail below
        s = np.ascontiguousarray(np.concatenate([s, A[rng.integers(0, len(A), 5)]])[:m + 3])
        n = len(s)
        for left in (0, 1):
            a = [C.c_int() for _ in range(3)]
            b = [C.c_int() for _ in range(3)]
            r0 = O.oracle_drop_sc(C.byref(P), s.ctypes.data, n, t.ctypes.data, m, left, *[C.byref(x) for x in a])
            r1 = emu.emu_drop_sc(C.byref(P), s.ctypes.data, n, t.ctypes.data, m, left, *[C.byref(x) for x in b])
            assert r0 == r1 and [x.value for x in a] == [x.value for x in b], (it, kind, left, n, m, p, bytes(s), bytes(t))


def test_skipped_leftover_extensions_are_dead_work(emu):
    """leftovers_matter() == false must imply that no outcome of the unpaired-chain extensions can change
    mr.type: leftover_type() is applied with mr_update_type (type only ever decreases), an extension can only
    lower a side's min_ret to CONCRD / CANDID / ORPHAN (chain_both_sides' return values) and flip its genic flag."""
    import itertools
    CONCRD, CANDID, ORPHAN = 0, 9, 11
    rets = (CONCRD, CANDID, ORPHAN)
    skipped = 0
    for T, m1, m2, can1, can2 in itertools.product(range(14), rets, rets, (0, 1), (0, 1)):
        # the reference only extends a side whose min_ret is not CONCRD yet
        c1, c2 = can1 and m1 != CONCRD, can2 and m2 != CONCRD
        if emu.emu_leftovers_matter(T, m1, int(c1), m2, int(c2)):
            continue
        skipped += 1
        f1 = {min(m1, e) for e in rets} if c1 else {m1}
        f2 = {min(m2, e) for e in rets} if c2 else {m2}
        for a, b, g1, g2 in itertools.product(f1, f2, (0, 1), (0, 1)):
            assert emu.emu_leftover_type(a, b, g1, g2) >= T, (T, m1, m2, can1, can2, a, b, g1, g2)
    assert skipped > 100


def _many_isoforms(ds, tmp_path, copies):
    """the data set's annotation with every transcript repeated `copies` times (new transcript ids, same exons)"""
    from conftest import _Shim
    out, block = [], []

    def flush():
        if block:
            for c in range(copies):
                out.extend(ln.replace('transcript_id "T', f'transcript_id "c{c}T') for ln in block)
            block.clear()

    for ln in ds.d.gtf_text.splitlines(keepends=True):
        kind = ln.split("\t")[2]
        if kind == "exon":
            block.append(ln)
            continue
        flush()
        if kind == "transcript":
            block.append(ln)
        else:
            out.append(ln)
    flush()
    gtf = str(tmp_path / "iso.gtf")
    open(gtf, "w").write("".join(out))
    hi = cl.HostIndex(ds.d.contigs, ds.d.chr_table, gtf, kmer=ds.kmer)
    sh = _Shim(ds, ds.batch)
    sh.hi = hi
    return sh


def test_more_than_64_common_transcripts(emu, ds_tiny, tmp_path):
    """Genes with hundreds of isoforms (Ensembl has them): |common_tid| of a mate pair exceeds the 64 entries the device
    code keeps per lane; the rest is re-derived from the two intervals in the same order, so results equal the oracle's
    unbounded vector (reference src/utils.cpp:322-354)."""
    sh = _many_isoforms(ds_tiny, tmp_path, 45)
    av = sh.hi.annots[0]
    ntid = np.diff(np.ctypeslib.as_array(av.seg_tid_off, shape=(av.n_seg + 1,)).astype(np.int64))
    assert ntid.max() > 2 * 64 and av.n_trans == 45 * ds_tiny.hi.annots[0].n_trans
    _emu_rounds(emu, sh, cl.default_params())
    _emu_rounds(emu, sh, cl.default_params(scan_level=2, max_ed=6))


def test_variety_annotation_all_rounds(emu, ds_variety):
    """nested / overlapping / opposite-strand / single-exon genes etc. (conftest.ds_variety): kernel bodies on the product's
    annotation vs the oracle on its own"""
    _emu_rounds(emu, ds_variety, cl.default_params())
    _emu_rounds(emu, ds_variety, cl.default_params(scan_level=2, max_ed=6))


def test_reads_of_21_seeds(emu24, ds_long, ds_dirty):
    """300-bp reads at k = 14 (21 seeds, beyond the 16 of the default build): the kernel bodies as the library's second build
    compiles them vs the oracle; plus ragged 250 - 300 bp reads"""
    _emu_rounds(emu24, ds_long, cl.default_params(kmer=14))
    d = ds_long.d
    rng = np.random.default_rng(5)
    l1, l2 = rng.integers(250, 301, d.seq1.shape[0]), rng.integers(250, 301, d.seq1.shape[0])
    s1 = np.concatenate([d.seq1[i, :l1[i]] for i in range(len(l1))])
    s2 = np.concatenate([d.seq2[i, :l2[i]] for i in range(len(l2))])
    from conftest import _Shim
    _emu_rounds(emu24, _Shim(ds_long, cl.ReadBatch(s1, s2, l1, l2)), cl.default_params(kmer=14, scan_level=1))
    _emu_rounds(emu24, ds_long, cl.default_params(kmer=14, seed_lim=60, max_ed=6))      # many seeds above the hit limit at k = 14


def tiny_exon_case(tmp_path, n_tiny, n_del, seed=3):
    """One gene whose single transcript runs E0 (100 bp), `n_tiny` exons of 10 bp (150-bp introns), E_last (200 bp); one read
    pair: R1 = 50 bases of E0, every tiny exon, the rest from E_last, with one base deleted from the middle of each of the
    first `n_del` tiny exons (after the first); R2 = reverse complement of E_last's tail.  The two seeds of R1 fall in E0, so
    the right extension walks 1 + n_tiny middle pieces (one memo key each) before its end piece, and with n_del > band that
    end piece has rlen < qlen -- the two conditions DESIGN.md's memo note is about."""
    from conftest import _Shim
    rng = np.random.default_rng(seed)
    A = np.frombuffer(b"ACGT", np.uint8)
    L = 30_000
    chrom = A[rng.integers(0, 4, L)].copy()
    exons, pos = [(5001, 5100)], 5100
    for _ in range(n_tiny):
        pos += 150
        exons.append((pos + 1, pos + 10))
        pos += 10
    pos += 150
    exons.append((pos + 1, pos + 200))
    t = synth.Transcript(0, "T0", "+", exons)
    g = synth.Gene(0, "G0", exons[0][0], exons[-1][1], "+", [t])
    gtf = str(tmp_path / f"tiny_exons_{n_tiny}_{n_del}.gtf")
    open(gtf, "w").write(synth.gtf_text([g], ["chr1"]))
    pieces = [chrom[5050:5100]]
    for k in range(n_tiny):
        s, e = exons[1 + k]
        ex = chrom[s - 1:e]
        pieces.append(np.delete(ex, 5) if 1 <= k <= n_del else ex)
    s, e = exons[-1]
    r1 = np.concatenate(pieces + [chrom[s - 1:e]])[:150]
    r2 = synth.revcomp(chrom[e - 150:e])
    contigs, table = synth.pack_genome(["chr1"], [chrom], 1 << 20)

    class D:
        pass
    ds = D()
    ds.d, ds.kmer = D(), 20
    ds.d.contigs, ds.d.chr_table = contigs, table
    ds.hi = cl.HostIndex(contigs, table, gtf, kmer=20)
    ds.ohi = op.OracleIndex(contigs, table, gtf, kmer=20)
    return _Shim(ds, cl.ReadBatch(r1[None, :].copy(), r2[None, :].copy()))


def test_extension_memo_limit(emu, tmp_path):
    """The device memo keeps 8 exon alignments per extend call (cm_core.h MEMO_N) where the reference's std::map is unbounded
    (src/extend.cpp:299,375).  A full table alone, or a short end piece alone, is exact; both together would let the
    reference serve a colliding key the device recomputes, so the device code reports CM_ELIMIT (ERR_MEMO) instead."""
    P = cl.default_params(max_ed=6)
    full_only = tiny_exon_case(tmp_path, 10, 0)           # 11 middle pieces: inserts dropped, no deletions
    _emu_rounds(emu, full_only, P)
    st, act = op.default_state(P, 1)
    assert op.map_round(P, full_only.ohi.views[0], full_only.ohi.annots[0], full_only.batch, True, st, act)[0] == 0     # CONCRD: the extension ran through
    short_end_only = tiny_exon_case(tmp_path, 6, 4)       # 7 middle pieces + end = 8 keys: nothing dropped, end piece 1 short
    _emu_rounds(emu, short_end_only, P)
    both = tiny_exon_case(tmp_path, 10, 4)
    st1, act1 = op.default_state(P, 1)
    cat1 = np.full(1, -1, np.int32)
    rc = emu.emu_map_round(C.byref(P), C.byref(both.hi.views[0]), C.byref(both.hi.annots[0]), C.byref(both.batch.c), 1, st1.ctypes.data,
                           act1.ctypes.data, cat1.ctypes.data)
    assert rc == 16                                       # cmc::ERR_MEMO and nothing else
    # ... which on the device queues the pair for the re-run with a spill area behind the 8 entries (cm_hot.hip RetryArgs): exact
    emu.emu_map_round_spill.argtypes = emu.emu_map_round.argtypes + [C.c_int]
    for cap in (3, 2040):                                 # 3: the spill area fills up too -> still flagged; 2040: the device's size
        st2, act2 = op.default_state(P, 1)
        rc = emu.emu_map_round_spill(C.byref(P), C.byref(both.hi.views[0]), C.byref(both.hi.annots[0]), C.byref(both.batch.c), 1, st2.ctypes.data,
                                     act2.ctypes.data, cat1.ctypes.data, cap)
        assert rc == (16 if cap == 3 else 0)
    st0, act0 = op.default_state(P, 1)
    cat0 = op.map_round(P, both.ohi.views[0], both.ohi.annots[0], both.batch, True, st0, act0)
    assert st2.tobytes() == st0.tobytes() and (act2 == act0).all() and cat1[0] == cat0[0]


def test_seed_touch_count(emu, ds_tiny):
    """cmc::seed_probe: occurrences, first entry and the search-touch counter (8 bytes per element the
    reference's two binary searches look at, src/match_read.cpp:54-110) against those two searches run literally on the index
    arrays, for k-mers that occur once, several times (repeat families) and not at all.  (At k = 20 a bucket holds 4 entries on
    average -- 4^14 buckets -- so a galloping upper bound, tried in round 2, has nothing to save.)"""
    P = cl.default_params()
    iv = ds_tiny.hi.views[0]
    n_ent = int(iv.n_entries)
    W = 14                                               # CM_WINDOW_SIZE: bases in the bucket hash
    off = np.ctypeslib.as_array(iv.bucket_off, ((1 << (2 * W)) + 1,))
    cks = np.ctypeslib.as_array(iv.checksum, (n_ent,))
    pos = np.ctypeslib.as_array(iv.pos, (n_ent,))
    g = ds_tiny.d.contigs[0]
    emu.emu_probe.argtypes = [C.POINTER(cl.Params), C.POINTER(cl.IndexView), C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    emu.emu_probe.restype = C.c_uint32
    code = np.full(256, -1, np.int64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i

    def reference(kmer):
        cd = code[kmer]
        if (cd < 0).any():
            return 0, 0, 0
        hv = 0
        for x in cd[:W]:
            hv = hv * 4 + int(x)
        cv = 0
        for x in cd[W:]:
            cv = cv * 4 + int(x)
        target = cv - 65536 if cv >= 32768 else cv
        b0, b1 = int(off[hv]), int(off[hv + 1])
        if b1 == b0:
            return 0, 0, 0
        it, t = cks[b0:b1].astype(np.int64), 0
        lb, ub = 1, b1 - b0
        while lb < ub:
            mid = (lb + ub) // 2
            t += 1
            if target <= it[mid - 1]:
                ub = mid
            else:
                lb = mid + 1
        t += 1
        if ub < lb or target != it[lb - 1]:
            return 0, 0, t
        LB, ub = lb, b1 - b0
        while lb < ub:
            mid = (lb + ub + 1) // 2
            t += 1
            if target < it[mid - 1]:
                ub = mid - 1
            else:
                lb = mid
        t += 1
        UB = lb if target == it[lb - 1] else LB
        return UB - LB + 1, b0 + LB - 1, t

    rng = np.random.default_rng(3)
    runs = np.diff(np.flatnonzero(np.concatenate(([True], cks[1:] != cks[:-1], [True]))))
    assert runs.max() >= 3                               # the repeat families give runs of equal checksums
    long_starts = np.flatnonzero(np.concatenate(([True], cks[1:] != cks[:-1])))[runs >= 2]
    where = list(rng.integers(0, len(g) - 20, 600)) + [int(pos[s]) - 1 for s in long_starts[-400:]]
    seen_multi = 0
    for p in where:
        kmer = g[p:p + 20].copy()
        for mutate in (False, True):
            if mutate:
                kmer[int(rng.integers(W, 20))] = ord("ACGT"[int(rng.integers(0, 4))])
            st, tc = C.c_uint32(0), C.c_uint32(0)
            raw = emu.emu_probe(C.byref(P), C.byref(iv), kmer.ctypes.data, 0, C.byref(st), C.byref(tc))
            want = reference(kmer)
            assert (raw, tc.value) == (want[0], want[2]) and (raw == 0 or st.value == want[1]), (p, mutate, raw, st.value, tc.value, want)
            seen_multi += raw >= 2
    assert seen_multi >= 100


@pytest.mark.parametrize("kmer", [20, 22, 15])
def test_bucket_descriptors(emu, tmp_path_factory, kmer):
    """The device answers a k-mer probe from a 16-byte bucket descriptor (offset, count, the checksums of a small bucket packed at
    2 (k - 14) bits each: cmc::desc_pack, built when a contig is loaded) instead of the offset table + checksum array.  Same
    code on the CPU with descriptors built by the same packer: seed ranges, occurrence counts and search-touch counts equal the
    array path's, and a whole mapping round equals the oracle's; k = 20 (12-bit checksums, 7 inline), k = 22 (16 bits, the int16
    quirk, 5 inline) and k = 15 (2 bits, 44 inline)."""
    from conftest import DataSet
    ds = DataSet(tmp_path_factory.mktemp(f"desc{kmer}"), "tiny", 500, 61 + kmer, kmer=kmer)
    P = cl.default_params(kmer=kmer)
    iv, b = ds.hi.views[0], ds.batch
    S = b.max_len() // kmer
    a0, b0, c0 = (np.zeros(b.n * 4 * S, np.uint32) for _ in range(3))
    assert emu.emu_seed_batch(C.byref(P), C.byref(iv), C.byref(b.c), S, a0.ctypes.data, b0.ctypes.data, c0.ctypes.data) == 0
    emu.emu_probe.argtypes = [C.POINTER(cl.Params), C.POINTER(cl.IndexView), C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    emu.emu_probe.restype = C.c_uint32
    g = ds.d.contigs[0]
    rng = np.random.default_rng(kmer)
    where = rng.integers(0, len(g) - kmer, 3000)

    def probes():
        out = []
        for p in where:
            st, tc = C.c_uint32(0), C.c_uint32(0)
            raw = emu.emu_probe(C.byref(P), C.byref(iv), g[p:p + kmer].ctypes.data, 0, C.byref(st), C.byref(tc))
            out.append((raw, st.value if raw else 0, tc.value))
        return out

    plain = probes()
    emu.emu_build_desc.argtypes = [C.POINTER(cl.Params), C.POINTER(cl.IndexView)]
    assert emu.emu_build_desc(C.byref(P), C.byref(iv)) == 0
    try:
        assert probes() == plain
        a1, b1, c1 = (np.zeros_like(a0) for _ in range(3))
        assert emu.emu_seed_batch(C.byref(P), C.byref(iv), C.byref(b.c), S, a1.ctypes.data, b1.ctypes.data, c1.ctypes.data) == 0
        assert (b0 == b1).all() and (c0 == c1).all() and (a0[c0 > 0] == a1[c0 > 0]).all()
        _emu_rounds(emu, ds, P)
    finally:
        emu.emu_free_desc()
    assert sum(r[0] > 0 for r in plain) > 1000 and max(r[0] for r in plain) >= 3      # runs that cross the 32- and 64-bit word borders
