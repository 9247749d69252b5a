"""Array-by-array comparison of two sets of host-side views (the product's builders against the oracle's own), shared by the
CPU suite, the GPU chr21 test and tests/diag/builders_1g.py."""
import ctypes as C
import hashlib

import numpy as np


def arr(p, n, dt=np.uint32, copy=True):
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (max(int(n), 1) * np.dtype(dt).itemsize,)).view(dt)[:int(n)]
    return a.copy() if copy else a


def assert_index_views_equal(a, b, digest=None):
    assert a.n_entries == b.n_entries and a.ref_len == b.ref_len, (a.n_entries, b.n_entries)
    for name, n, dt in (("bucket_off", 2 ** 28 + 1, np.uint32), ("checksum", a.n_entries, np.uint16), ("pos", a.n_entries, np.uint32)):
        x, y = arr(getattr(a, name), n, dt, copy=False), arr(getattr(b, name), n, dt, copy=False)
        assert np.array_equal(x, y), name
        if digest is not None:
            digest[name] = hashlib.sha256(x.tobytes()).hexdigest()


def assert_annot_views_equal(x, y, digest=None):
    h = hashlib.sha256()
    for f in ("n_iv", "n_seg", "n_trans", "n_gene", "n_bits", "n_chr", "n_giv"):
        assert getattr(x, f) == getattr(y, f), f
    for f, n in (("iv_spos", "n_iv"), ("iv_epos", "n_iv"), ("iv_max_end", "n_iv"), ("iv_min_end", "n_iv"), ("iv_max_next_exon", "n_iv"),
                 ("seg_start", "n_seg"), ("seg_end", "n_seg"), ("seg_next_exon_beg", "n_seg"), ("seg_gene_id", "n_seg"), ("gene_start", "n_gene"),
                 ("gene_end", "n_gene"), ("chr_shift", "n_chr"), ("giv_spos", "n_giv"), ("giv_epos", "n_giv")):
        u = arr(getattr(x, f), getattr(x, n))
        assert np.array_equal(u, arr(getattr(y, f), getattr(y, n))), f
        h.update(u.tobytes())
    for off, val, n in (("iv_seg_off", "iv_seg", "n_iv"), ("seg_tid_off", "seg_tid", "n_seg"), ("giv_gene_off", "giv_gene", "n_giv"), ("t2s_off", "t2s", "n_trans")):
        ox, oy = arr(getattr(x, off), getattr(x, n) + 1), arr(getattr(y, off), getattr(y, n) + 1)
        assert np.array_equal(ox, oy), off
        dt = np.uint8 if val == "t2s" else np.uint32
        u = arr(getattr(x, val), ox[-1], dt)
        assert np.array_equal(u, arr(getattr(y, val), oy[-1], dt)), val
        h.update(ox.tobytes())
        h.update(u.tobytes())
    assert np.array_equal(arr(x.trans_start_ind, x.n_trans, np.int32), arr(y.trans_start_ind, y.n_trans, np.int32))
    assert np.array_equal(arr(x.chr_id, x.n_chr, np.int32), arr(y.chr_id, y.n_chr, np.int32))
    for f in ("near_border_bits", "intronic_bits"):
        u = arr(getattr(x, f), x.n_bits // 64, np.uint64, copy=False)
        assert np.array_equal(u, arr(getattr(y, f), y.n_bits // 64, np.uint64, copy=False)), f
        h.update(u.tobytes())
    if digest is not None:
        digest["annotation"] = h.hexdigest()


def assert_host_views_equal(hi, ohi, digests=None):
    """every array of every contig's cm_index_view and cm_annot_view"""
    assert hi.n_contigs == ohi.n_contigs
    for ci in range(hi.n_contigs):
        d = {} if digests is not None else None
        assert_index_views_equal(hi.views[ci], ohi.views[ci], d)
        assert_annot_views_equal(hi.annots[ci], ohi.annots[ci], d)
        if digests is not None:
            digests.append(d)
