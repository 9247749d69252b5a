"""Host-side index / annotation builders against independent numpy / pure-Python restatements of the
reference's build rules (mrsfast HashTable.c:769-839; gene_annotation.cpp:191-399;
interval_tree_impl.h:40-127,186-242)."""
import ctypes as C
import os

import numpy as np
import pytest

from circminer_amd import lib as cl


def _np(ptr, n, dt):
    return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].astype(dt)


def test_index_matches_numpy_restatement(ds_tiny):
    iv = ds_tiny.hi.views[0]
    g = ds_tiny.d.contigs[0]
    k, W = 20, 14
    code = np.full(256, 4, np.int64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    c = code[g]
    n = len(g) - k + 1
    bad = np.convolve((c == 4).astype(np.int64), np.ones(k, np.int64), "valid")
    starts = np.nonzero(bad == 0)[0]
    val = np.zeros(len(starts), np.uint64)
    for j in range(k):
        val = (val << np.uint64(2)) | c[starts + j].astype(np.uint64)
    order = np.lexsort((starts, val))                    # (k-mer value, position) == (bucket, checksum, pos)
    val, pos = val[order], starts[order] + 1
    assert iv.n_entries == len(pos)
    assert (_np(iv.pos, iv.n_entries, np.int64) == pos).all()
    assert (_np(iv.checksum, iv.n_entries, np.uint64) == (val & np.uint64((1 << (2 * (k - W))) - 1))).all()
    off = _np(iv.bucket_off, 4 ** W + 1, np.int64)
    bucket = (val >> np.uint64(2 * (k - W))).astype(np.int64)
    assert (off[1:] - off[:-1] == np.bincount(bucket, minlength=4 ** W)).all()
    assert n > 0


def _ref_intervals(segs):
    """FlatIntervalTree::build restated literally in Python (insertion procedure)."""
    iv = []          # [spos, epos, [seg idx...]]

    def handle(cur, si):
        s, e = segs[si][0], segs[si][1]
        m = iv[cur]
        if m[0] < s:
            pre = m[1]
            m[1] = s - 1
            iv.insert(cur + 1, [s, min(pre, e), m[2] + [si]])
            if pre < e:
                return cur + 2, True
            if pre == e:
                return cur, False
            iv.insert(cur + 2, [e + 1, pre, list(m[2])])
            return cur, False
        if m[1] < e:
            m[2].append(si)
            return cur + 1, True
        if m[1] == e:
            m[2].append(si)
            return cur, False
        iv.insert(cur, [m[0], e, m[2] + [si]])
        iv[cur + 1][0] = e + 1
        return cur, False

    j = 0
    for si, (s, e, *_r) in enumerate(segs):
        while j < len(iv) and s > iv[j][1]:
            j += 1
        if j == len(iv):
            iv.append([s, e, [si]])
        else:
            cur, rem = j, False
            while cur < len(iv):
                cur, rem = handle(cur, si)
                if not rem:
                    break
            if cur == len(iv) and rem:
                iv.append([iv[cur - 1][1] + 1, e, [si]])
    return iv


def test_annotation_matches_python_restatement(ds_tiny2r):
    d = ds_tiny2r.d
    for con in range(ds_tiny2r.hi.n_contigs):
        av = ds_tiny2r.hi.annots[con]
        shift = {name: (cid - 1, st) for name, cid, st, _ in d.chr_table}
        # unique segments keyed like UniqSeg::operator< (start, end, gene, -next)
        segmap, n_gene, n_trans = {}, 0, 0
        genes = []
        for g in d.genes:
            c, sh = shift[d.chr_names[g.chrom]]
            if c != con:
                continue
            gid = n_gene
            n_gene += 1
            genes.append((g.start + sh, g.end + sh))
            for t in g.transcripts:
                tid = n_trans
                n_trans += 1
                ex = t.exons
                for k2, (a, b) in enumerate(ex):
                    nxt = ex[k2 + 1][0] + sh if k2 + 1 < len(ex) else 0
                    segmap.setdefault((a + sh, b + sh, gid, -nxt), []).append(tid)
        keys = sorted(segmap)
        segs = [(a, b, g_, -n_) for (a, b, g_, n_) in keys]
        iv = _ref_intervals(segs)
        assert av.n_iv == len(iv) and av.n_seg == len(segs) and av.n_gene == n_gene and av.n_trans == n_trans
        assert (_np(av.iv_spos, av.n_iv, np.int64) == [x[0] for x in iv]).all()
        assert (_np(av.iv_epos, av.n_iv, np.int64) == [x[1] for x in iv]).all()
        off = _np(av.iv_seg_off, av.n_iv + 1, np.int64)
        flat = _np(av.iv_seg, off[-1], np.int64)
        assert list(flat) == [s for x in iv for s in x[2]]
        assert (_np(av.seg_start, av.n_seg, np.int64) == [s[0] for s in segs]).all()
        assert (_np(av.seg_next_exon_beg, av.n_seg, np.int64) == [s[3] for s in segs]).all()
        assert (_np(av.seg_gene_id, av.n_seg, np.int64) == [s[2] for s in segs]).all()
        toff = _np(av.seg_tid_off, av.n_seg + 1, np.int64)
        tids = _np(av.seg_tid, toff[-1], np.int64)
        assert list(tids) == [t for k2 in keys for t in segmap[k2]]
        assert (_np(av.gene_start, av.n_gene, np.int64) == [x[0] for x in genes]).all()
        # per-interval caches and the transcript -> segment table
        mx = [max(segs[s][1] for s in x[2]) for x in iv]
        assert (_np(av.iv_max_end, av.n_iv, np.int64) == mx).all()
        starts = _np(av.trans_start_ind, av.n_trans, np.int64)
        t2o = _np(av.t2s_off, av.n_trans + 1, np.int64)
        t2s = _np(av.t2s, t2o[-1], np.int64)
        for i, x in enumerate(iv):
            for s in x[2]:
                state = 1 if x[0] == segs[s][0] else (3 if x[1] == segs[s][1] else 2)
                for t in segmap[keys[s]]:
                    assert starts[t] <= i and t2s[t2o[t] + i - starts[t]] == state
        # bitsets: near-border flanks and intronic positions
        nb = np.unpackbits(_np(av.near_border_bits, av.n_bits // 64, np.uint64).view(np.uint8), bitorder="little")
        it = np.unpackbits(_np(av.intronic_bits, av.n_bits // 64, np.uint64).view(np.uint8), bitorder="little")
        exp_nb = np.zeros_like(nb)
        exp_it = np.zeros_like(it)
        for a, b in genes:
            exp_it[a:b + 1] = 1
        for (a, b, _g, _n) in segs:
            exp_it[a:b + 1] = 0
            if a >= 300:
                exp_nb[a - 300:a] = 1
            if b + 1 >= 300:
                exp_nb[b - 299:b + 1] = 1
        assert (nb == exp_nb).all() and (it == exp_it).all()


def _annot_bytes(av):
    """every array of a cm_annot_view as bytes, keyed by field name"""
    n = {"iv_spos": av.n_iv, "iv_epos": av.n_iv, "iv_max_end": av.n_iv, "iv_min_end": av.n_iv, "iv_max_next_exon": av.n_iv,
         "iv_seg_off": av.n_iv + 1, "seg_start": av.n_seg, "seg_end": av.n_seg, "seg_next_exon_beg": av.n_seg, "seg_gene_id": av.n_seg,
         "seg_tid_off": av.n_seg + 1, "trans_start_ind": av.n_trans, "t2s_off": av.n_trans + 1, "gene_start": av.n_gene,
         "gene_end": av.n_gene, "near_border_bits": av.n_bits // 64, "intronic_bits": av.n_bits // 64, "chr_shift": av.n_chr,
         "chr_id": av.n_chr, "iv_bucket": av.n_iv_bucket}
    out = {k: np.ctypeslib.as_array(getattr(av, k), shape=(max(v, 1),))[:v].tobytes() for k, v in n.items()}
    out["iv_seg"] = np.ctypeslib.as_array(av.iv_seg, shape=(max(int(av.iv_seg_off[av.n_iv]), 1),))[:int(av.iv_seg_off[av.n_iv])].tobytes()
    out["seg_tid"] = np.ctypeslib.as_array(av.seg_tid, shape=(max(int(av.seg_tid_off[av.n_seg]), 1),))[:int(av.seg_tid_off[av.n_seg])].tobytes()
    out["t2s"] = np.ctypeslib.as_array(av.t2s, shape=(max(int(av.t2s_off[av.n_trans]), 1),))[:int(av.t2s_off[av.n_trans])].tobytes()
    out["counts"] = (av.n_iv, av.n_seg, av.n_trans, av.n_gene, av.n_bits, av.n_chr)
    return out


def test_gtf_line_handling(ds_tiny2r, tmp_path):
    """Rows the reference's load_gtf ignores or reads leniently (gene_annotation.cpp:79-143,200-215): comments, other feature
    types, unknown chromosomes, runs of tabs (tokenize drops empty fields), short rows; and a long gene crossing many
    bitset words.  The annotation must be identical to the one built from the clean file."""
    d = ds_tiny2r.d
    clean = d.gtf_text.splitlines(keepends=True)
    rng = np.random.default_rng(2)
    messy = ["#!genome-build synthetic\n"]
    for ln in clean:
        f = ln.split("\t")
        r = rng.random()
        if r < 0.2:
            messy.append("# a comment\n")
        elif r < 0.4:
            messy.append("\t".join([f[0], f[1], "CDS"] + f[3:]))
        elif r < 0.5:
            messy.append("\t".join(["0"] + f[1:]))               # chromosome that is not in the table
        elif r < 0.6:
            messy.append("chr_short\tx\n")
        if rng.random() < 0.3:
            ln = "\t\t".join(f[:4]) + "\t\t\t" + "\t".join(f[4:])   # empty fields collapse
        messy.append(ln)
    p = tmp_path / "messy.gtf"
    p.write_text("".join(messy))
    hi = cl.HostIndex(d.contigs, d.chr_table, str(p), kmer=20)
    for con in range(hi.n_contigs):
        a, b = _annot_bytes(ds_tiny2r.hi.annots[con]), _annot_bytes(hi.annots[con])
        for k in a:
            assert a[k] == b[k], (con, k)


def test_gene_interval_table_and_overlap_query(ds_tiny2r, tmp_path):
    """genes_int_map of stage 2 (GTFParser::get_gene_overlap, src/gene_annotation.cpp:243-256,362-365,572-585): the interval
    construction over the gene spans, genes with identical spans folded into the first, and the point query.  The synthetic
    genes rarely overlap, so a second annotation with nested, overlapping, duplicated and abutting genes is built as well."""
    L = cl.load()

    def check(av, genes):                     # genes: [(start, end)] in file order, contig coordinates
        first = {}
        for gid, (a, b) in enumerate(genes):
            first.setdefault((a, b), gid)
        keys = sorted(first)
        iv = _ref_intervals([(a, b) for a, b in keys])
        assert av.n_giv == len(iv)
        assert (_np(av.giv_spos, av.n_giv, np.int64) == [x[0] for x in iv]).all()
        assert (_np(av.giv_epos, av.n_giv, np.int64) == [x[1] for x in iv]).all()
        off = _np(av.giv_gene_off, av.n_giv + 1, np.int64)
        flat = _np(av.giv_gene, off[-1], np.int64)
        assert list(flat) == [first[keys[s]] for x in iv for s in x[2]]
        # point query against brute force over the kept genes
        rng = np.random.default_rng(1)
        pts = set(int(p) for p in rng.integers(0, max(b for _a, b in genes) + 50, 400))
        for a, b in genes:
            pts.update((a - 1, a, b, b + 1))
        for p in sorted(x for x in pts if x >= 0):
            g, n = cl.u32p(), C.c_uint32(0)
            assert L.cm_host_gene_overlap(C.byref(av), p, C.byref(g), C.byref(n)) == 0
            got = sorted(g[i] for i in range(n.value))
            want = sorted(first[k] for k in keys if k[0] <= p <= k[1])
            assert got == want, (p, got, want)

    d = ds_tiny2r.d
    shift = {name: (cid - 1, st) for name, cid, st, _ in d.chr_table}
    for con in range(ds_tiny2r.hi.n_contigs):
        genes = [(g.start + shift[d.chr_names[g.chrom]][1], g.end + shift[d.chr_names[g.chrom]][1]) for g in d.genes
                 if shift[d.chr_names[g.chrom]][0] == con]
        check(ds_tiny2r.hi.annots[con], genes)
    spans = [(100, 900), (100, 900), (300, 500), (450, 1200), (901, 950), (1200, 1300), (2000, 2100), (2050, 2060), (2050, 2060), (3000, 3000)]
    gtf = tmp_path / "genes.gtf"
    with open(gtf, "w") as f:
        for i, (a, b) in enumerate(spans):
            f.write(f'chrA\tx\tgene\t{a}\t{b}\t.\t+\t.\tgene_id "g{i}";\n')
            f.write(f'chrA\tx\ttranscript\t{a}\t{b}\t.\t+\t.\tgene_id "g{i}"; transcript_id "t{i}";\n')
            f.write(f'chrA\tx\texon\t{a}\t{b}\t.\t+\t.\tgene_id "g{i}"; transcript_id "t{i}";\n')
    chrs = cl.chr_array([("chrA", 1, 0, 5000), ("chrB", 2, 0, 4000)])
    lens = (C.c_uint32 * 2)(5000, 4000)
    out = (cl.AnnotView * 2)()
    assert L.cm_host_build_annotation(str(gtf).encode(), chrs, 2, lens, 2, 300, out) == 0
    check(out[0], spans)
    assert out[1].n_giv == 1 and out[1].giv_spos[0] == 0xFFFFFFFF        # dummy interval of an annotation-less contig
    g, n = cl.u32p(), C.c_uint32(7)
    assert L.cm_host_gene_overlap(C.byref(out[1]), 123, C.byref(g), C.byref(n)) == 0 and n.value == 0
    L.cm_host_free_annotation(out, 2)


def _arr(p, n, dt=np.uint32):
    import ctypes as C
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (max(int(n), 1) * np.dtype(dt).itemsize,)).view(dt)[:int(n)].copy()


@pytest.mark.parametrize("name", ["ds_tiny", "ds_tiny2r", "ds_small", "ds_variety"])
def test_product_builders_equal_the_oracles_own(name, request):
    """host_index.cpp / host_annot.cpp (product) against oracle/cm_oracle_build.cpp (the oracle's own builders, written from the
    reference's HashTable.c, gene_annotation.cpp and interval_tree_impl.h with the reference's containers): every array of every
    contig's cm_index_view and cm_annot_view is identical -- including the variety preset with nested / overlapping /
    opposite-strand / single-exon / duplicate-span genes, an exon next to the chromosome start and unordered gene blocks."""
    from builders_util import assert_host_views_equal
    ds = request.getfixturevalue(name)
    assert ds.ohi is not ds.hi
    assert_host_views_equal(ds.hi, ds.ohi)
    if name == "ds_variety":        # the shapes are really there
        av = ds.hi.annots[0]
        nseg = np.diff(_arr(av.iv_seg_off, av.n_iv + 1))
        assert nseg.max() >= 3 and (np.diff(_arr(av.giv_gene_off, av.n_giv + 1)) >= 2).any()          # overlapping exons, overlapping genes
        assert _arr(av.gene_start, av.n_gene).min() < 300                                          # the gene next to the chromosome start


def test_index_shared_between_ranks_through_files(ds_tiny2r, tmp_path):
    """bench.py with N > 1 ranks: rank 0 builds the index and dumps the arrays, the other ranks memory-map them
    (HostIndex.save_index / index_dir): same views"""
    ds = ds_tiny2r
    d = str(tmp_path / "shared")
    ds.hi.save_index(d)
    h2 = cl.HostIndex(ds.d.contigs, ds.d.chr_table, ds.gtf, kmer=ds.kmer, index_dir=d)
    for a, b in zip(ds.hi.views, h2.views):
        assert a.n_entries == b.n_entries and a.ref_len == b.ref_len and a.contig_num == b.contig_num
        assert np.array_equal(_arr(a.bucket_off, 2 ** 28 + 1), _arr(b.bucket_off, 2 ** 28 + 1))
        assert np.array_equal(_arr(a.checksum, a.n_entries, np.uint16), _arr(b.checksum, b.n_entries, np.uint16))
        assert np.array_equal(_arr(a.pos, a.n_entries), _arr(b.pos, b.n_entries))
    h2.close()
    open(os.path.join(d, "c0.pos"), "ab").write(b"\0\0\0\0")           # a torn file is refused
    with pytest.raises(RuntimeError):
        cl.HostIndex(ds.d.contigs, ds.d.chr_table, ds.gtf, kmer=ds.kmer, index_dir=d)


def test_index_hit_stats_equal_a_kmer_count(ds_small):
    """cm_host_index_stats (positions whose k-mer occurs more than once / more than seedLim times in the contig: the two
    fractions SURVEY 8(d) fixes for the hg38-like preset) against a plain count of the contig's 20-mers."""
    import collections
    g = ds_small.d.contigs[0].tobytes()
    k = 20
    cnt = collections.Counter(g[i:i + k] for i in range(len(g) - k + 1) if set(g[i:i + k]) <= set(b"ACGT"))
    n = sum(cnt.values())
    for lim in (1, 3, 500):
        want = (n, sum(c for c in cnt.values() if c > 1), sum(c for c in cnt.values() if c > lim), len(cnt))
        assert ds_small.hi.hit_stats(lim, 3)[0] == want
    assert want[1] > 0                                    # the preset has repeat families
