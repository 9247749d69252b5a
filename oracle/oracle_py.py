"""ctypes binding of the CPU oracle (oracle/_build/libcmoracle.so).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under circminer_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from circminer_amd import lib as cl

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libcmoracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(HERE, "cm_oracle.cpp"), os.path.join(HERE, "cm_oracle_build.cpp"), os.path.join(HERE, "..", "include", "circminer_hot.h")]
    if (not force and os.path.exists(SO) and os.path.getmtime(SO) >= max(os.path.getmtime(x) for x in srcs)):
        return SO
    subprocess.check_call(["make", "-C", HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return SO


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        L = C.CDLL(SO)
        vp, pp = C.c_void_p, C.POINTER
        L.oracle_seed_batch.restype = C.c_int
        L.oracle_seed_batch.argtypes = [pp(cl.Params), pp(cl.IndexView), pp(cl.Reads), C.c_uint32, vp, vp, vp]
        L.oracle_chain_batch.restype = C.c_int
        L.oracle_chain_batch.argtypes = [pp(cl.Params), pp(cl.IndexView), pp(cl.AnnotView), pp(cl.Reads), vp, vp, vp]
        L.oracle_map_round.restype = C.c_int
        L.oracle_map_round.argtypes = [pp(cl.Params), pp(cl.IndexView), pp(cl.AnnotView), pp(cl.Reads), C.c_int, vp, vp, vp,
                                       C.c_uint64, C.c_uint64]
        L.oracle_default_state.restype = None
        L.oracle_default_state.argtypes = [pp(cl.Params), vp, vp, C.c_uint64]
        L.oracle_edit_side.restype = C.c_int
        L.oracle_edit_side.argtypes = [pp(cl.Params), vp, C.c_int, vp, C.c_int, C.c_int, pp(C.c_int), pp(C.c_int)]
        L.oracle_drop_sc.restype = C.c_int
        L.oracle_drop_sc.argtypes = [pp(cl.Params), vp, C.c_int, vp, C.c_int, C.c_int, pp(C.c_int), pp(C.c_int), pp(C.c_int)]
        L.oracle_one_side.restype = C.c_int
        L.oracle_one_side.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]
        _lib = L
    return _lib


def seeds(params, iv, batch, n_slots):
    L = load()
    k = batch.n * 4 * n_slots
    a, b, c = (np.zeros(max(k, 1), np.uint32) for _ in range(3))
    rc = L.oracle_seed_batch(C.byref(params), C.byref(iv), C.byref(batch.c), n_slots, a.ctypes.data, b.ctypes.data, c.ctypes.data)
    assert rc == 0
    return a[:k], b[:k], c[:k]


def chains(params, iv, av, batch):
    L = load()
    ch = np.zeros(max(batch.n, 1) * 4 * cl.CM_BESTCHAINLIM, dtype=cl.CHAIN_DTYPE)
    nc = np.zeros(max(batch.n, 1) * 4, np.int32)
    hh = np.zeros(max(batch.n, 1) * 4, np.int32)
    rc = L.oracle_chain_batch(C.byref(params), C.byref(iv), C.byref(av), C.byref(batch.c), ch.ctypes.data, nc.ctypes.data, hh.ctypes.data)
    assert rc == 0
    return ch[:batch.n * 4 * cl.CM_BESTCHAINLIM], nc[:batch.n * 4], hh[:batch.n * 4]


def default_state(params, n):
    L = load()
    st = np.zeros(max(n, 1), dtype=cl.MAPPED_DTYPE)
    act = np.zeros(max(n, 1), dtype=np.uint8)
    L.oracle_default_state(C.byref(params), st.ctypes.data, act.ctypes.data, n)
    return st[:n], act[:n]


def map_round(params, iv, av, batch, is_last, state, active, p0=0, p1=None):
    L = load()
    cat = np.full(max(batch.n, 1), -1, dtype=np.int32)
    rc = L.oracle_map_round(C.byref(params), C.byref(iv), C.byref(av), C.byref(batch.c), int(is_last), state.ctypes.data,
                            active.ctypes.data, cat.ctypes.data, p0, batch.n if p1 is None else p1)
    assert rc == 0, rc
    return cat[:batch.n]


def map_all_rounds(params, host_index, batch):
    """Every round of `mapping()` (reference src/circminer.cpp:229-308) on the CPU oracle."""
    st, act = default_state(params, batch.n)
    cats = []
    for ci in range(host_index.n_contigs):
        last = ci == host_index.n_contigs - 1
        cats.append(map_round(params, host_index.views[ci], host_index.annots[ci], batch, last, st, act))
    return st, act, cats


def map_all_rounds_mt(params, host_index, batch, n_threads=None, p1=None):
    """map_all_rounds on several host threads (process_read is a pure function of the pair; ctypes releases the GIL):
    pairs [0, p1) only when p1 is given.  Returns (state, active, category of the last round)."""
    import threading
    L = load()
    n = batch.n if p1 is None else min(int(p1), batch.n)
    T = max(1, n_threads or os.cpu_count() or 1)
    st, act = default_state(params, batch.n)
    cat = np.full(max(batch.n, 1), -1, dtype=np.int32)

    def work(a, b):
        for ci in range(host_index.n_contigs):
            rc = L.oracle_map_round(C.byref(params), C.byref(host_index.views[ci]), C.byref(host_index.annots[ci]), C.byref(batch.c),
                                    int(ci == host_index.n_contigs - 1), st.ctypes.data, act.ctypes.data, cat.ctypes.data, a, b)
            assert rc == 0, rc

    th = [threading.Thread(target=work, args=((n * i) // T, (n * (i + 1)) // T)) for i in range(T)]
    [t.start() for t in th]
    [t.join() for t in th]
    return st, act, cat[:batch.n]


def circ_run(params, host_index, chr_table, names, batch, states, candid_path, report_path, window=8):
    """Stage 2 (ProcessCirc::do_process) on records given in the order of the sorted remain files: names[i], batch pair i and
    states[i] (the MatchedRead carried in R1's header, chromosome coordinates).  Writes candidates.pam and circ_report."""
    L = load()
    L.oracle_circ_run.restype = C.c_int
    n_con = host_index.n_contigs
    views = (cl.IndexView * n_con)(*host_index.views)
    keep = [n.encode() if isinstance(n, str) else n for n in names]
    name_arr = (C.c_char_p * max(len(keep), 1))(*keep)
    chr_keep = [t[0].encode() for t in chr_table]
    chr_arr = (C.c_char_p * len(chr_keep))(*chr_keep)
    chr_contig = np.asarray([t[1] - 1 for t in chr_table], dtype=np.uint32)
    chr_shift = np.asarray([t[2] for t in chr_table], dtype=np.uint32)
    st = np.ascontiguousarray(states)
    rc = L.oracle_circ_run(C.byref(params), C.c_int(window), C.c_uint32(n_con), views, host_index.annots, C.c_uint32(len(chr_table)), chr_arr,
                           C.c_void_p(chr_contig.ctypes.data), C.c_void_p(chr_shift.ctypes.data), C.c_uint64(len(keep)), name_arr,
                           C.byref(batch.c), C.c_void_p(st.ctypes.data), candid_path.encode(), report_path.encode())
    assert rc == 0, rc


class OracleIndex:
    """Index + annotation of a packed genome built by the ORACLE's own builders (oracle/cm_oracle_build.cpp, written from the
    reference's HashTable.c / gene_annotation.cpp / interval_tree_impl.h), same attributes as circminer_amd.lib.HostIndex.
    The parity tests feed the oracle from this and the HIP path from the product's builders."""

    def __init__(self, contigs, chr_table, gtf_path, kmer=20, max_read_len=300):
        L = load()
        self.L = L
        self.contigs = [np.ascontiguousarray(c, dtype=np.uint8) for c in contigs]
        self.views = []
        for ci, g in enumerate(self.contigs):
            iv = cl.IndexView()
            rc = L.oracle_build_index(C.c_void_p(g.ctypes.data), C.c_uint32(len(g)), C.c_int32(kmer), C.c_int32(ci), C.byref(iv))
            assert rc == 0, rc
            self.views.append(iv)
        n_con = len(self.contigs)
        self._names = [t[0].encode() for t in chr_table]
        chrs = (cl.ChrInfo * len(chr_table))(*[cl.ChrInfo(self._names[i], t[1], t[2], t[3]) for i, t in enumerate(chr_table)])
        clen = np.asarray([len(c) for c in self.contigs], dtype=np.uint32)
        self.annots = (cl.AnnotView * n_con)()
        self._holders = (C.c_void_p * n_con)()
        rc = L.oracle_build_annotation(gtf_path.encode(), chrs, C.c_uint32(len(chr_table)), C.c_void_p(clen.ctypes.data), C.c_uint32(n_con),
                                       C.c_int32(max_read_len), self.annots, self._holders)
        assert rc == 0, rc
        self.chr_table = list(chr_table)
        self.n_contigs = n_con
        self.kmer = kmer

    def close(self):
        if self.views:
            for iv in self.views:
                self.L.oracle_free_index(C.byref(iv))
            self.L.oracle_free_annotation(self._holders, C.c_uint32(self.n_contigs))
            self.views = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
