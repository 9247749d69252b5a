"""ctypes mirror of include/circminer_hot.h and the loader of the product library.

The product library (``circminer_amd/csrc/libcmhot.so``) is hand-written HIP for gfx950 behind a
C-ABI; this module only declares the structs / prototypes and never computes anything itself.
There is no CPU fallback: if the library is missing or no HIP device is present the calls fail.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CM_LIB") or os.path.join(HERE, "csrc", "libcmhot.so")

CM_BESTCHAINLIM = 30
CM_MAX_CHAIN_FRAGS = 16
CM_CONTIG_SIZE = 1_100_000_000

CAT = dict(CONCRD=0, DISCRD=1, CHIORF=2, CHIBSJ=3, CHI2BSJ=4, CONGEN=5, CHIFUS=6, CONGNM=7, OEA2=8,
           CANDID=9, OEANCH=10, ORPHAN=11, NOPROC_MANYHIT=12, NOPROC_NOMATCH=13)

u8p, u16p, u32p, u64p, i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_int32))


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kmer", "seed_lim", "max_read_len", "scan_level", "max_ed", "max_sc", "band",
                                         "max_tlen", "max_intron", "max_chain_len", "device", "reserved")]


class CircRes(C.Structure):
    _fields_ = [("chr", C.c_char_p), ("rname", C.c_char_p), ("spos", C.c_uint32), ("epos", C.c_uint32), ("type", C.c_int32),
                ("reserved", C.c_int32), ("start_signal", C.c_char_p), ("end_signal", C.c_char_p), ("start_bp_ref", C.c_char_p),
                ("end_bp_ref", C.c_char_p)]


class MappingArgs(C.Structure):
    _fields_ = [("index_path", C.c_char_p), ("index_info_path", C.c_char_p), ("gtf_path", C.c_char_p), ("fastq1", C.c_char_p),
                ("fastq2", C.c_char_p), ("out_prefix", C.c_char_p), ("params", Params), ("report", C.c_int32), ("n_threads", C.c_int32),
                ("batch_pairs", C.c_uint64), ("rank", C.c_int32), ("world", C.c_int32)]


class CircStats(C.Structure):
    _fields_ = [("pairs", C.c_uint64), ("candidate_rows", C.c_uint64), ("calls", C.c_uint64), ("seconds", C.c_double)]


class MappingStats(C.Structure):
    _fields_ = [("pairs", C.c_uint64), ("bsj_pairs", C.c_uint64), ("by_type", C.c_uint64 * 14), ("rounds", C.c_int32),
                ("reserved", C.c_int32), ("seconds_load", C.c_double), ("seconds_map", C.c_double), ("seconds_parse", C.c_double),
                ("seconds_device", C.c_double), ("seconds_write", C.c_double)]


def default_params(**kw) -> Params:
    """Defaults of reference src/commandline_parser.cpp:7-33 / src/common.h:39-53."""
    d = dict(kmer=20, seed_lim=500, max_read_len=300, scan_level=0, max_ed=4, max_sc=7, band=3, max_tlen=500,
             max_intron=2_000_000, max_chain_len=30, device=0, reserved=0)
    d.update(kw)
    return Params(**d)


class CircArgs(C.Structure):
    _fields_ = [("index_path", C.c_char_p), ("index_info_path", C.c_char_p), ("gtf_path", C.c_char_p), ("out_prefix", C.c_char_p),
                ("params", Params), ("last_round", C.c_int32), ("window_size", C.c_int32), ("n_threads", C.c_int32), ("reserved", C.c_int32)]


class IndexView(C.Structure):
    _fields_ = [("contig_num", C.c_int32), ("ref_len", C.c_uint32), ("genome", u8p), ("bucket_off", u32p),
                ("checksum", u16p), ("pos", u32p), ("n_entries", C.c_uint64)]


class IndexRaw(C.Structure):
    _fields_ = [("contig_num", C.c_int32), ("ref_len", C.c_uint32), ("genome", u8p), ("n_buckets", C.c_uint32), ("hv", u32p),
                ("count14", u32p), ("table", C.c_void_p), ("table_slots", C.c_uint64)]


class AnnotView(C.Structure):
    _fields_ = [("n_iv", C.c_uint32), ("iv_spos", u32p), ("iv_epos", u32p), ("iv_max_end", u32p), ("iv_min_end", u32p),
                ("iv_max_next_exon", u32p), ("iv_seg_off", u32p), ("iv_seg", u32p),
                ("n_seg", C.c_uint32), ("seg_start", u32p), ("seg_end", u32p), ("seg_next_exon_beg", u32p),
                ("seg_gene_id", u32p), ("seg_tid_off", u32p), ("seg_tid", u32p),
                ("n_trans", C.c_uint32), ("trans_start_ind", i32p), ("t2s_off", u32p), ("t2s", u8p),
                ("n_gene", C.c_uint32), ("gene_start", u32p), ("gene_end", u32p),
                ("n_bits", C.c_uint64), ("near_border_bits", u64p), ("intronic_bits", u64p),
                ("n_chr", C.c_uint32), ("chr_shift", u32p), ("chr_id", i32p),
                ("iv_bucket", u32p), ("iv_bucket_shift", C.c_uint32), ("n_iv_bucket", C.c_uint32),
                ("n_giv", C.c_uint32), ("giv_spos", u32p), ("giv_epos", u32p), ("giv_gene_off", u32p), ("giv_gene", u32p)]


class MappedRead(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("spos_r1", "spos_r2", "epos_r1", "epos_r2", "qspos_r1", "qspos_r2",
                                          "qepos_r1", "qepos_r2", "mlen_r1", "mlen_r2")] + \
               [(n, C.c_int32) for n in ("ed_r1", "ed_r2", "type", "tlen", "contig_num", "chr_id")] + \
               [("junc_num", C.c_uint16), ("r1_forward", C.c_uint8), ("r2_forward", C.c_uint8),
                ("gm_compatible", C.c_uint8), ("pad", C.c_uint8 * 3)]


MAPPED_DTYPE = np.dtype([(n, "<u4") for n in ("spos_r1", "spos_r2", "epos_r1", "epos_r2", "qspos_r1", "qspos_r2",
                                               "qepos_r1", "qepos_r2", "mlen_r1", "mlen_r2")] +
                        [(n, "<i4") for n in ("ed_r1", "ed_r2", "type", "tlen", "contig_num", "chr_id")] +
                        [("junc_num", "<u2"), ("r1_forward", "u1"), ("r2_forward", "u1"), ("gm_compatible", "u1"),
                         ("pad", "u1", (3,))])
assert MAPPED_DTYPE.itemsize == C.sizeof(MappedRead) == 72
RECORD_DTYPE = np.dtype([("pair", "<u8"), ("state", MAPPED_DTYPE)])      # cm_record, 80 bytes

CHAIN_DTYPE = np.dtype([("score", "<f4"), ("chain_len", "<u4"), ("rpos", "<u4", (CM_MAX_CHAIN_FRAGS,)),
                        ("qpos", "<i4", (CM_MAX_CHAIN_FRAGS,))])
assert CHAIN_DTYPE.itemsize == 136


class Reads(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("seq1", u8p), ("off1", u64p), ("seq2", u8p), ("off2", u64p)]


class ChrInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("contig_id", C.c_uint32), ("start_pos", C.c_uint32), ("len", C.c_uint32)]


class FastqBatch(C.Structure):
    _fields_ = [("reads", Reads), ("qual1", u8p), ("qual2", u8p), ("names1", C.c_void_p), ("names2", C.c_void_p),
                ("name_off1", u64p), ("name_off2", u64p), ("prior", C.c_void_p)]


def ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


class ReadBatch:
    """Host-side concatenated reads (keeps the numpy buffers alive)."""

    def __init__(self, seq1: np.ndarray, seq2: np.ndarray, len1=None, len2=None):
        if seq1.ndim == 2:
            n, L1 = seq1.shape
            L2 = seq2.shape[1]
            self.off1 = (np.arange(n + 1, dtype=np.uint64) * L1)
            self.off2 = (np.arange(n + 1, dtype=np.uint64) * L2)
            self.seq1 = np.ascontiguousarray(seq1, dtype=np.uint8).reshape(-1)
            self.seq2 = np.ascontiguousarray(seq2, dtype=np.uint8).reshape(-1)
        else:
            self.seq1 = np.ascontiguousarray(seq1, dtype=np.uint8)
            self.seq2 = np.ascontiguousarray(seq2, dtype=np.uint8)
            self.off1 = np.concatenate([[0], np.cumsum(len1)]).astype(np.uint64)
            self.off2 = np.concatenate([[0], np.cumsum(len2)]).astype(np.uint64)
            n = len(self.off1) - 1
        if self.seq1.size == 0:
            self.seq1 = np.zeros(1, np.uint8)
        if self.seq2.size == 0:
            self.seq2 = np.zeros(1, np.uint8)
        self.n = int(n)
        self.c = Reads(self.n, ptr(self.seq1, u8p), ptr(self.off1, u64p), ptr(self.seq2, u8p), ptr(self.off2, u64p))

    def max_len(self) -> int:
        if self.n == 0:
            return 0
        return int(max(np.diff(self.off1).max(), np.diff(self.off2).max()))


_lib = None


RELEASE_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint64)      # cm_fastq_set_release_hook: (user, ptr, bytes)


def load(path: str = LIB_PATH) -> C.CDLL:
    """Load libcmhot.so and declare every prototype of include/circminer_hot.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the hot path is HIP-only; there is no CPU fallback)")
    L = C.CDLL(path)
    vp = C.c_void_p
    pp = C.POINTER
    sigs = {
        "cm_create": (C.c_int, [pp(Params), pp(vp)]),
        "cm_destroy": (None, [vp]),
        "cm_last_error": (C.c_char_p, [vp]),
        "cm_load_contig": (C.c_int, [vp, C.c_int, pp(IndexView)]),
        "cm_load_annotation": (C.c_int, [vp, C.c_int, pp(AnnotView)]),
        "cm_load_contig_raw": (C.c_int, [vp, C.c_int, pp(IndexRaw)]),
        "cm_host_next_contig_raw": (C.c_int, [vp, C.c_int, pp(IndexRaw), pp(C.c_int)]),
        "cm_unload_contig": (C.c_int, [vp, C.c_int]),
        "cm_reads_upload": (C.c_int, [vp, pp(Reads), vp]),
        "cm_reads_stage": (C.c_int, [vp, pp(Reads), vp]),
        "cm_reads_swap": (C.c_int, [vp]),
        "cm_map_round": (C.c_int, [vp, C.c_int, C.c_int]),
        "cm_map_rounds": (C.c_int, [vp, i32p, C.c_int, C.c_int]),
        "cm_reads_download": (C.c_int, [vp, vp, vp, vp]),
        "cm_map_batch": (C.c_int, [vp, C.c_int, C.c_int, pp(Reads), vp, vp, vp]),
        "cm_sync": (C.c_int, [vp]),
        "cm_reads_reset": (C.c_int, [vp]),
        "cm_collect_active": (C.c_int, [vp, C.c_uint64, vp, vp, pp(C.c_uint64)]),
        "cm_collect_records": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, pp(C.c_uint64)]),
        "cm_collect_records_device": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, pp(C.c_uint64)]),
        "cm_host_alloc": (C.c_int, [vp, C.c_uint64, pp(vp)]),
        "cm_host_free": (C.c_int, [vp, vp]),
        "cm_host_register": (C.c_int, [vp, vp, C.c_uint64]),
        "cm_host_unregister": (C.c_int, [vp, vp]),
        "cm_type_histogram": (C.c_int, [vp, pp(C.c_uint64)]),
        "cm_seed_batch": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_uint32, pp(C.c_uint32)]),
        "cm_chain_batch": (C.c_int, [vp, C.c_int, vp, vp, vp]),
        "cm_prof_enable": (C.c_int, [vp, C.c_int]),
        "cm_prof_reset": (C.c_int, [vp]),
        "cm_prof_get": (C.c_int, [vp, pp(C.c_double), pp(C.c_uint64)]),
        "cm_prof_counters": (C.c_int, [vp, pp(C.c_uint64)]),
        "cm_host_build_index": (C.c_int, [u8p, C.c_uint32, C.c_int32, C.c_int32, C.c_int, pp(IndexView)]),
        "cm_host_free_index": (None, [pp(IndexView)]),
        "cm_host_index_stats": (C.c_int, [pp(IndexView), C.c_int32, C.c_int, pp(C.c_uint64)]),
        "cm_host_build_annotation": (C.c_int, [C.c_char_p, pp(ChrInfo), C.c_uint32, u32p, C.c_uint32, C.c_int32,
                                               pp(AnnotView)]),
        "cm_host_free_annotation": (None, [pp(AnnotView), C.c_uint32]),
        "cm_host_pack_genome": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32]),
        "cm_host_read_index_info": (C.c_int, [C.c_char_p, pp(pp(ChrInfo)), pp(C.c_uint32)]),
        "cm_host_free_index_info": (None, [pp(ChrInfo), C.c_uint32]),
        "cm_host_write_index": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_int, C.c_int]),
        "cm_host_open_index": (C.c_int, [C.c_char_p, pp(vp), pp(C.c_int32), pp(C.c_int32), pp(C.c_uint32)]),
        "cm_host_next_contig": (C.c_int, [vp, C.c_int, pp(IndexView), pp(C.c_int)]),
        "cm_host_next_contig_genome": (C.c_int, [vp, pp(IndexView), pp(C.c_int)]),
        "cm_host_free_loaded_contig": (None, [pp(IndexView)]),
        "cm_host_close_index": (None, [vp]),
        "cm_fastq_open": (C.c_int, [C.c_char_p, C.c_char_p, pp(ChrInfo), C.c_uint32, C.c_int32, pp(vp)]),
        "cm_fastq_open_shard": (C.c_int, [C.c_char_p, C.c_char_p, pp(ChrInfo), C.c_uint32, C.c_int32, C.c_int32, C.c_int32, C.c_int, pp(vp),
                                          pp(C.c_uint64), pp(C.c_uint64)]),
        "cm_merge_parts": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32]),
        "cm_fastq_next": (C.c_int, [vp, C.c_uint64, pp(FastqBatch)]),
        "cm_fastq_close": (None, [vp]),
        "cm_fastq_set_release_hook": (None, [vp, RELEASE_HOOK, vp]),
        "cm_writer_open": (C.c_int, [C.c_char_p, C.c_char_p, pp(ChrInfo), C.c_uint32, pp(vp)]),
        "cm_write_remain": (C.c_int, [vp, pp(FastqBatch), vp, vp, C.c_uint64]),
        "cm_write_remain_records": (C.c_int, [vp, pp(FastqBatch), vp, C.c_uint64]),
        "cm_write_pam": (C.c_int, [vp, pp(FastqBatch), vp, vp, C.c_uint64]),
        "cm_host_gene_overlap": (C.c_int, [pp(AnnotView), C.c_uint32, pp(u32p), pp(C.c_uint32)]),
        "cm_sort_remain": (C.c_int, [C.c_char_p, C.c_char_p]),
        "cm_regional_table_build": (C.c_int, [vp, C.c_uint32, C.c_int32, C.c_int32, pp(u32p), pp(u32p)]),
        "cm_regional_table_free": (None, [u32p, u32p]),
        "cm_circ_report": (C.c_int, [pp(CircRes), C.c_uint64, C.c_char_p]),
        "cm_mapping_run": (C.c_int, [pp(MappingArgs), pp(MappingStats), C.c_char_p, C.c_uint64]),
        "cm_circ_call": (C.c_int, [pp(Params), C.c_int32, C.c_uint32, pp(IndexView), pp(AnnotView), pp(ChrInfo), C.c_uint32, pp(FastqBatch),
                                   C.c_char_p, C.c_char_p, pp(CircStats)]),
        "cm_circ_run": (C.c_int, [pp(CircArgs), pp(CircStats), C.c_char_p, C.c_uint64]),
        "cm_write_sam_header": (C.c_int, [vp]),
        "cm_write_sam": (C.c_int, [vp, pp(FastqBatch), vp, vp, C.c_uint64]),
        "cm_writer_flush": (C.c_int, [vp]),
        "cm_writer_close": (None, [vp]),
    }
    sigs["cm_abi_sizes"] = (C.c_int, [pp(C.c_uint32), C.c_uint32])
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    # the ctypes mirrors below must be the structs the library was compiled with
    got = (C.c_uint32 * 16)()
    n = L.cm_abi_sizes(got, 16)
    mine = [C.sizeof(t) for t in (Params, IndexView, AnnotView, MappedRead, Reads, C.c_uint8 * RECORD_DTYPE.itemsize, ChrInfo, FastqBatch, MappingArgs,
                                  MappingStats, CircRes, CircArgs, CircStats, IndexRaw)]
    if n != len(mine) or list(got[:n]) != mine:
        raise RuntimeError(f"circminer_amd.lib: struct sizes differ from {path}: library {list(got[:max(n, 0)])}, ctypes {mine}")
    _lib = L
    return L


EXPORTED_SYMBOLS = ["cm_create", "cm_destroy", "cm_last_error", "cm_load_contig", "cm_load_contig_raw", "cm_host_next_contig_raw", "cm_load_annotation",
                    "cm_unload_contig", "cm_reads_upload", "cm_reads_stage", "cm_reads_swap", "cm_map_round", "cm_map_rounds", "cm_reads_download", "cm_map_batch",
                    "cm_sync", "cm_reads_reset", "cm_collect_active", "cm_collect_records", "cm_collect_records_device", "cm_abi_sizes", "cm_host_alloc", "cm_host_free", "cm_host_register", "cm_host_unregister", "cm_type_histogram", "cm_write_remain_records", "cm_seed_batch", "cm_chain_batch", "cm_prof_enable", "cm_prof_reset", "cm_prof_get",
                    "cm_prof_counters", "cm_host_build_index", "cm_host_free_index", "cm_host_index_stats", "cm_host_build_annotation",
                    "cm_host_free_annotation", "cm_host_pack_genome", "cm_host_read_index_info", "cm_host_free_index_info",
                    "cm_host_write_index", "cm_host_open_index", "cm_host_next_contig", "cm_host_next_contig_genome", "cm_host_free_loaded_contig",
                    "cm_host_close_index", "cm_fastq_open", "cm_fastq_open_shard", "cm_merge_parts", "cm_fastq_next", "cm_fastq_set_release_hook", "cm_fastq_close", "cm_writer_open", "cm_write_remain",
                    "cm_write_pam", "cm_write_sam_header", "cm_write_sam", "cm_writer_flush", "cm_writer_close", "cm_mapping_run", "cm_sort_remain", "cm_circ_report", "cm_circ_call", "cm_circ_run", "cm_host_gene_overlap", "cm_regional_table_build", "cm_regional_table_free"]


class HostIndex:
    """Owns the host-side index + annotation of a packed genome (built by the C++ host builders).

    index_dir: take the k-mer index arrays from files written by save_index() (memory-mapped, shared between the rank
    processes of a node) instead of building them; the annotation is always built here (seconds)."""

    def __init__(self, contigs, chr_table, gtf_path, kmer=20, max_read_len=300, n_threads=8, index_dir=None):
        L = load()
        self.L = L
        self.contigs = [np.ascontiguousarray(c, dtype=np.uint8) for c in contigs]
        self.views = []
        self._mapped = []
        self.kmer = kmer
        for ci, g in enumerate(self.contigs):
            iv = IndexView()
            if index_dir is None:
                rc = L.cm_host_build_index(ptr(g, u8p), len(g), kmer, ci, n_threads, C.byref(iv))
                if rc != 0:
                    raise RuntimeError(f"cm_host_build_index failed: {rc}")
            else:
                arrs = [np.memmap(os.path.join(index_dir, f"c{ci}.{nm}"), dtype=dt, mode="r")
                        for nm, dt in (("off", np.uint32), ("cks", np.uint16), ("pos", np.uint32))]
                if len(arrs[0]) != (1 << 28) + 1 or len(arrs[1]) != len(arrs[2]) or int(arrs[0][-1]) != len(arrs[2]):
                    raise RuntimeError(f"index files of contig {ci} in {index_dir} are inconsistent")
                self._mapped.append(arrs)
                iv = IndexView(ci, len(g), ptr(g, u8p), C.cast(arrs[0].ctypes.data, u32p), C.cast(arrs[1].ctypes.data, u16p),
                               C.cast(arrs[2].ctypes.data, u32p), len(arrs[2]))
            self.views.append(iv)
        n_con = len(self.contigs)
        self._names = [t[0].encode() for t in chr_table]
        chrs = (ChrInfo * len(chr_table))(*[ChrInfo(self._names[i], t[1], t[2], t[3]) for i, t in enumerate(chr_table)])
        clen = np.asarray([len(c) for c in self.contigs], dtype=np.uint32)
        self.annots = (AnnotView * n_con)()
        rc = L.cm_host_build_annotation(gtf_path.encode(), chrs, len(chr_table), ptr(clen, u32p), n_con, max_read_len,
                                        self.annots)
        if rc != 0:
            raise RuntimeError(f"cm_host_build_annotation failed: {rc}")
        self.chr_table = list(chr_table)
        self.n_contigs = n_con

    def hit_stats(self, seed_lim=500, n_threads=8):
        """Per contig (indexed positions, positions whose k-mer occurs more than once, positions whose k-mer occurs more than
        seed_lim times, distinct k-mers): the repeat content a probe of this index sees (cm_host_index_stats)."""
        res = []
        for iv in self.views:
            o = (C.c_uint64 * 4)()
            rc = self.L.cm_host_index_stats(C.byref(iv), seed_lim, n_threads, o)
            if rc != 0:
                raise RuntimeError(f"cm_host_index_stats failed: {rc}")
            res.append(tuple(int(x) for x in o))
        return res

    def save_index(self, index_dir):
        """Raw dumps of the flattened index arrays (bucket offsets, checksums, positions) of every contig."""
        os.makedirs(index_dir, exist_ok=True)
        for ci, iv in enumerate(self.views):
            n = int(iv.n_entries)
            np.ctypeslib.as_array(iv.bucket_off, ((1 << 28) + 1,)).tofile(os.path.join(index_dir, f"c{ci}.off"))
            np.ctypeslib.as_array(iv.checksum, (max(n, 1),))[:n].tofile(os.path.join(index_dir, f"c{ci}.cks"))
            np.ctypeslib.as_array(iv.pos, (max(n, 1),))[:n].tofile(os.path.join(index_dir, f"c{ci}.pos"))

    def close(self):
        if self.views:
            if not self._mapped:
                for iv in self.views:
                    self.L.cm_host_free_index(C.byref(iv))
            self._mapped = []
            self.L.cm_host_free_annotation(self.annots, self.n_contigs)
            self.views = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_genome(fasta: str, contig_size: int = CM_CONTIG_SIZE):
    """FASTA -> <fasta>.packed.fa + <fasta>.packed.fa.index.info (stock CircMiner file names); returns both paths."""
    packed, info = fasta + ".packed.fa", fasta + ".packed.fa.index.info"
    rc = load().cm_host_pack_genome(fasta.encode(), packed.encode(), info.encode(), contig_size)
    if rc != 0:
        raise RuntimeError(f"cm_host_pack_genome failed ({rc})")
    return packed, info


def write_index(packed_fa: str, kmer: int = 20, compact: bool = False, n_threads: int = 8) -> str:
    """packed FASTA -> <packed_fa>.index in the mrsfast file format stock CircMiner reads."""
    idx = packed_fa + ".index"
    rc = load().cm_host_write_index(packed_fa.encode(), idx.encode(), kmer, int(compact), n_threads)
    if rc != 0:
        raise RuntimeError(f"cm_host_write_index failed ({rc})")
    return idx


class IndexFile:
    """Iterates the packed contigs of a stock index file as IndexView objects (one mapping round each).
    The view handed out is valid until the next iteration step / close()."""

    def __init__(self, path: str, n_threads: int = 8, genome_only: bool = False, raw: bool = False):
        self.L = load()
        self.h = C.c_void_p()
        self.genome_only = genome_only          # cm_host_next_contig_genome: the k-mer tables are stepped over (what stage 2 does)
        self.raw = raw                          # cm_host_next_contig_raw: records as in the file, for HotPath.load_contig_raw
        kmer, full, nrec = C.c_int32(0), C.c_int32(0), C.c_uint32(0)
        rc = self.L.cm_host_open_index(path.encode(), C.byref(self.h), C.byref(kmer), C.byref(full), C.byref(nrec))
        if rc != 0:
            raise RuntimeError(f"cm_host_open_index failed ({rc}): {path}")
        self.kmer, self.full, self.n_records, self.n_threads = kmer.value, bool(full.value), nrec.value, n_threads
        self._cur = None

    def __iter__(self):
        return self

    def _drop(self):
        if self._cur is not None:
            self.L.cm_host_free_loaded_contig(C.byref(self._cur))
            self._cur = None

    def __next__(self):
        self._drop()
        iv, loaded = IndexView(), C.c_int(0)
        if self.raw:
            rw = IndexRaw()
            rc = self.L.cm_host_next_contig_raw(self.h, self.n_threads, C.byref(rw), C.byref(loaded))
            if rc != 0:
                raise RuntimeError(f"cm_host_next_contig_raw failed ({rc})")
            if not loaded.value:
                raise StopIteration
            return rw                          # arrays belong to the file handle (valid until the call after next)
        if self.genome_only:
            rc = self.L.cm_host_next_contig_genome(self.h, C.byref(iv), C.byref(loaded))
        else:
            rc = self.L.cm_host_next_contig(self.h, self.n_threads, C.byref(iv), C.byref(loaded))
        if rc != 0:
            raise RuntimeError(f"cm_host_next_contig failed ({rc})")
        if not loaded.value:
            raise StopIteration
        self._cur = iv
        return iv

    def close(self):
        self._drop()
        if self.h:
            self.L.cm_host_close_index(self.h)
            self.h = C.c_void_p()


def chr_array(chr_table):
    """ChrInfo array from rows (name, contig_id (1-based), start_pos, len) -- the rows of .index.info."""
    arr = (ChrInfo * max(len(chr_table), 1))()
    for i, (name, contig, start, ln) in enumerate(chr_table):
        arr[i] = ChrInfo(name.encode() if isinstance(name, str) else name, contig, start, ln)
    return arr


class ParsedBatch:
    """One batch from FastqReader: `.c`/`.n` like ReadBatch (goes to HotPath.upload), `.prior` = carried states or None."""

    def __init__(self, fb: FastqBatch):
        self.fb = fb
        self.c = fb.reads
        self.n = int(fb.reads.n_pairs)
        self.prior = None
        if fb.prior:
            self.prior = np.ctypeslib.as_array(C.cast(fb.prior, C.POINTER(C.c_uint8)), (self.n * MAPPED_DTYPE.itemsize,)).view(MAPPED_DTYPE).copy()

    def max_len(self) -> int:
        o1 = np.ctypeslib.as_array(self.c.off1, (self.n + 1,))
        o2 = np.ctypeslib.as_array(self.c.off2, (self.n + 1,))
        return int(max(np.diff(o1).max(initial=0), np.diff(o2).max(initial=0)))

    def name(self, i: int, mate: int = 1) -> str:
        base, off = (self.fb.names1, self.fb.name_off1) if mate == 1 else (self.fb.names2, self.fb.name_off2)
        return C.string_at(base + off[i]).decode()

    def seq(self, i: int, mate: int = 1) -> bytes:
        s, o = (self.c.seq1, self.c.off1) if mate == 1 else (self.c.seq2, self.c.off2)
        return bytes(np.ctypeslib.as_array(s, (int(o[self.n]),))[int(o[i]):int(o[i + 1])])

    def qual(self, i: int, mate: int = 1) -> bytes:
        q, o = (self.fb.qual1, self.c.off1) if mate == 1 else (self.fb.qual2, self.c.off2)
        return bytes(np.ctypeslib.as_array(q, (int(o[self.n]),))[int(o[i]):int(o[i + 1])])


def sort_remain(path: str, out: str = None) -> str:
    """<remain>.fastq -> <remain>.fastq.srt, ordered as ProcessCirc::sort_fq's GNU sort pipeline orders it (C locale)."""
    out = out or path + ".srt"
    rc = load().cm_sort_remain(path.encode(), out.encode())
    if rc != 0:
        raise RuntimeError(f"cm_sort_remain failed ({rc})")
    return out


def circ_report(calls, path: str):
    """calls: iterable of (chr, rname, spos, epos, type, start_signal, end_signal, start_bp_ref, end_bp_ref) -> .circ_report"""
    calls = list(calls)
    keep = [[x.encode() if isinstance(x, str) else x for x in c] for c in calls]
    arr = (CircRes * max(len(keep), 1))()
    for i, c in enumerate(keep):
        arr[i] = CircRes(c[0], c[1], c[2], c[3], c[4], 0, c[5], c[6], c[7], c[8])
    rc = load().cm_circ_report(arr, len(keep), path.encode())
    if rc != 0:
        raise RuntimeError(f"cm_circ_report failed ({rc})")


def circ_call(params, host_index, chr_table, sorted_batch, candidates_path, report_path, window=0):
    """cm_circ_call on an in-memory index: sorted_batch = the ParsedBatch read from the sorted remain files."""
    L = load()
    n_con = host_index.n_contigs
    views = (IndexView * n_con)(*host_index.views)
    chrs = chr_array(list(chr_table))
    st = CircStats()
    rc = L.cm_circ_call(C.byref(params), window, n_con, views, host_index.annots, chrs, len(chr_table), C.byref(sorted_batch.fb),
                        candidates_path.encode(), report_path.encode(), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"cm_circ_call failed ({rc})")
    return st


def run_circ(index_path, gtf, out_prefix, last_round, params=None, n_threads=4, index_info=None, window=0):
    """cm_circ_run: stage 2 from files to files (the reference's circ_detect(), src/circminer.cpp:347-352)."""
    L = load()
    a = CircArgs(index_path.encode(), (index_info or index_path + ".info").encode(), gtf.encode(), out_prefix.encode(),
                 params if params is not None else default_params(kmer=0), last_round, window, n_threads, 0)
    st = CircStats()
    err = C.create_string_buffer(1024)
    rc = L.cm_circ_run(C.byref(a), C.byref(st), err, len(err))
    if rc != 0:
        raise RuntimeError(f"cm_circ_run failed ({rc}): {err.value.decode()}")
    return st


def run_mapping(index_path, gtf, fastq1, fastq2, out_prefix, params=None, report=1, n_threads=4, batch_pairs=0, index_info=None, rank=0, world=1):
    """cm_mapping_run: stage 1 from files to files (the reference's mapping(), src/circminer.cpp:98-352).  rank / world: this
    process maps the rank-th block of pairs and writes .part<rank> files (merge_parts() on rank 0 afterwards)."""
    L = load()
    a = MappingArgs(index_path.encode(), (index_info or index_path + ".info").encode(), gtf.encode(), fastq1.encode(), fastq2.encode(),
                    out_prefix.encode(), params if params is not None else default_params(), report, n_threads, batch_pairs, rank, world)
    st = MappingStats()
    err = C.create_string_buffer(1024)
    rc = L.cm_mapping_run(C.byref(a), C.byref(st), err, len(err))
    if rc != 0:
        raise RuntimeError(f"cm_mapping_run failed ({rc}): {err.value.decode()}")
    return st


def merge_parts(out_prefix, rounds, world, report=1):
    """cm_merge_parts: the .part<rank> files of a sharded stage 1 -> the files one process would have written."""
    rc = load().cm_merge_parts(out_prefix.encode(), rounds, world, report)
    if rc != 0:
        raise RuntimeError(f"cm_merge_parts failed ({rc})")


class FastqReader:
    """Paired FASTQ (plain / gzip) -> batches in the cm_reads layout; a batch is valid until the next one is read.
    rank / world: only the rank-th contiguous block of pairs (cm_fastq_open_shard; plain-text files)."""

    def __init__(self, r1: str, r2: str, chr_table=(), max_ed: int = 4, rank: int = 0, world: int = 1, n_threads: int = 0):
        self.L = load()
        self._chrs = chr_array(list(chr_table))
        self.h = C.c_void_p()
        first, count = C.c_uint64(0), C.c_uint64(0)
        if world == 1:
            rc = self.L.cm_fastq_open(r1.encode(), r2.encode(), self._chrs, len(chr_table), max_ed, C.byref(self.h))
        else:
            rc = self.L.cm_fastq_open_shard(r1.encode(), r2.encode(), self._chrs, len(chr_table), max_ed, rank, world, n_threads, C.byref(self.h),
                                            C.byref(first), C.byref(count))
        if rc != 0:
            raise RuntimeError(f"cm_fastq_open failed ({rc})")
        self.first_pair, self.n_pairs = first.value, (count.value if world > 1 else None)

    def next_batch(self, max_pairs: int):
        fb = FastqBatch()
        rc = self.L.cm_fastq_next(self.h, max_pairs, C.byref(fb))
        if rc != 0:
            raise RuntimeError(f"cm_fastq_next failed ({rc}): malformed FASTQ")
        return ParsedBatch(fb) if fb.reads.n_pairs else None

    def close(self):
        if self.h:
            self.L.cm_fastq_close(self.h)
            self.h = C.c_void_p()


class RecordWriter:
    """PAM (path2=None) or remain-FASTQ pair writer in the reference's formats."""

    def __init__(self, path1: str, path2=None, chr_table=()):
        self.L = load()
        self._chrs = chr_array(list(chr_table))
        self.h = C.c_void_p()
        rc = self.L.cm_writer_open(path1.encode(), path2.encode() if path2 else None, self._chrs, len(chr_table), C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"cm_writer_open failed ({rc})")

    def _call(self, fn, batch: ParsedBatch, states: np.ndarray, sel):
        st = np.ascontiguousarray(states)
        sp, ns = None, 0
        if sel is not None:
            sel = np.ascontiguousarray(sel, dtype=np.uint64)
            sp, ns = sel.ctypes.data, len(sel)
        rc = fn(self.h, C.byref(batch.fb), st.ctypes.data, sp, ns)
        if rc != 0:
            raise RuntimeError(f"writer failed ({rc})")

    def write_remain(self, batch, states, sel=None):
        self._call(self.L.cm_write_remain, batch, states, sel)

    def write_pam(self, batch, states, sel=None):
        self._call(self.L.cm_write_pam, batch, states, sel)

    def write_sam_header(self):
        if self.L.cm_write_sam_header(self.h) != 0:
            raise RuntimeError("cm_write_sam_header failed")

    def write_sam(self, batch, states, sel=None):
        self._call(self.L.cm_write_sam, batch, states, sel)

    def close(self):
        if self.h:
            rc = self.L.cm_writer_flush(self.h)
            self.L.cm_writer_close(self.h)
            self.h = C.c_void_p()
            if rc != 0:
                raise RuntimeError(f"writing failed ({rc}): output file is incomplete")


class HotPath:
    """Thin OO wrapper over the cm_* C-ABI (one context per GPU)."""

    def __init__(self, params: Params):
        self.L = load()
        self.params = params
        self.h = C.c_void_p()
        rc = self.L.cm_create(C.byref(params), C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"cm_create failed ({rc}): no usable HIP device / bad params")
        self.n = 0
        self._pinned = []

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.L.cm_last_error(self.h).decode()}")

    def load_contig(self, slot, iv: IndexView, av: AnnotView = None):
        self._chk(self.L.cm_load_contig(self.h, slot, C.byref(iv)), "cm_load_contig")
        if av is not None:
            self._chk(self.L.cm_load_annotation(self.h, slot, C.byref(av)), "cm_load_annotation")

    def load_contig_raw(self, slot, raw: IndexRaw, av: AnnotView = None):
        """a record of IndexFile(raw=True): the table is flattened on the device"""
        self._chk(self.L.cm_load_contig_raw(self.h, slot, C.byref(raw)), "cm_load_contig_raw")
        if av is not None:
            self._chk(self.L.cm_load_annotation(self.h, slot, C.byref(av)), "cm_load_annotation")

    def upload(self, batch: ReadBatch, prior: np.ndarray = None):
        self.n = batch.n
        p = prior.ctypes.data if prior is not None else None
        self._chk(self.L.cm_reads_upload(self.h, C.byref(batch.c), p), "cm_reads_upload")

    def stage(self, batch: ReadBatch, prior: np.ndarray = None):
        """Start copying the next batch (asynchronous for page-locked arrays, see pinned_batch) while the resident one maps."""
        self._staged_n = batch.n
        p = prior.ctypes.data if prior is not None else None
        self._chk(self.L.cm_reads_stage(self.h, C.byref(batch.c), p), "cm_reads_stage")

    def swap(self):
        self._chk(self.L.cm_reads_swap(self.h), "cm_reads_swap")
        self.n = self._staged_n

    def pinned_batch(self, seq1: np.ndarray, seq2: np.ndarray) -> "ReadBatch":
        """A ReadBatch whose arrays live in page-locked memory of this context (copies of (n, L) uint8 matrices)."""
        n, L1 = seq1.shape
        L2 = seq2.shape[1]
        s1 = self.host_array(n * L1, np.uint8)
        s2 = self.host_array(n * L2, np.uint8)
        s1[:] = seq1.reshape(-1)
        s2[:] = seq2.reshape(-1)
        o1 = self.host_array(n + 1, np.uint64)
        o2 = self.host_array(n + 1, np.uint64)
        o1[:] = np.arange(n + 1, dtype=np.uint64) * L1
        o2[:] = np.arange(n + 1, dtype=np.uint64) * L2
        b = ReadBatch.__new__(ReadBatch)
        b.seq1, b.seq2, b.off1, b.off2, b.n = s1, s2, o1, o2, int(n)
        b.c = Reads(b.n, ptr(s1, u8p), ptr(o1, u64p), ptr(s2, u8p), ptr(o2, u64p))
        return b

    def map_round(self, slot, is_last):
        self._chk(self.L.cm_map_round(self.h, slot, int(is_last)), "cm_map_round")

    def map_rounds(self, slots, last_is_final=True):
        """All of `slots` in order in one call (round r + 1 is seeded / chained while round r's pair stage runs)."""
        arr = (C.c_int32 * len(slots))(*slots)
        self._chk(self.L.cm_map_rounds(self.h, arr, len(slots), int(last_is_final)), "cm_map_rounds")

    def sync(self):
        self._chk(self.L.cm_sync(self.h), "cm_sync")

    def reset(self):
        self._chk(self.L.cm_reads_reset(self.h), "cm_reads_reset")

    def collect_active(self, cap=None):
        """Active pairs (ascending index) and their carried state, as views of page-locked buffers
        that the next call overwrites (copy them to keep them)."""
        cap = int(cap if cap is not None else max(self.n, 1))
        if getattr(self, "_col_cap", 0) < cap:
            self._col_idx = self.host_array(cap, np.uint64)
            self._col_st = self.host_array(cap, MAPPED_DTYPE)
            self._col_cap = cap
        n = C.c_uint64(0)
        self._chk(self.L.cm_collect_active(self.h, cap, self._col_idx.ctypes.data, self._col_st.ctypes.data, C.byref(n)), "cm_collect_active")
        return self._col_idx[:n.value], self._col_st[:n.value]     # views: valid until the next call

    def collect_records(self, index_base=0, cap=None):
        """Active pairs as (global pair index, state) records -- dist.REC_DTYPE -- assembled on the device; a view of a
        page-locked buffer that the next call overwrites."""
        cap = int(cap if cap is not None else max(self.n, 1))
        if getattr(self, "_rec_cap", 0) < cap:
            self._rec = self.host_array(cap, RECORD_DTYPE)
            self._rec_cap = cap
        n = C.c_uint64(0)
        self._chk(self.L.cm_collect_records(self.h, int(index_base), cap, self._rec.ctypes.data, C.byref(n)), "cm_collect_records")
        return self._rec[:n.value]

    def collect_records_device(self, index_base, cap, dev_ptr):
        """Same records written to caller-owned device memory (`dev_ptr`: cap * 80 bytes on this context's GPU, e.g. a
        torch tensor's data_ptr()); returns the record count.  Used by dist.BsjGather."""
        n = C.c_uint64(0)
        self._chk(self.L.cm_collect_records_device(self.h, int(index_base), int(cap), C.c_void_p(int(dev_ptr)), C.byref(n)),
                  "cm_collect_records_device")
        return int(n.value)

    def host_array(self, n, dtype):
        """numpy array over page-locked memory from cm_host_alloc (freed with the context)."""
        dt = np.dtype(dtype)
        p = C.c_void_p()
        self._chk(self.L.cm_host_alloc(self.h, max(int(n), 1) * dt.itemsize, C.byref(p)), "cm_host_alloc")
        self._pinned.append(p.value)
        buf = (C.c_uint8 * (max(int(n), 1) * dt.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt, count=int(n))

    def download(self):
        st = np.zeros(self.n, dtype=MAPPED_DTYPE)
        cat = np.zeros(self.n, dtype=np.int32)
        act = np.zeros(self.n, dtype=np.uint8)
        self._chk(self.L.cm_reads_download(self.h, st.ctypes.data, cat.ctypes.data, act.ctypes.data), "cm_reads_download")
        return st, cat, act

    def seeds(self, slot, n_slots_cap=24):
        cap = self.n * 4 * n_slots_cap
        a = np.zeros(cap, np.uint32)
        b = np.zeros(cap, np.uint32)
        c = np.zeros(cap, np.uint32)
        ns = C.c_uint32(0)
        self._chk(self.L.cm_seed_batch(self.h, slot, a.ctypes.data, b.ctypes.data, c.ctypes.data, cap, C.byref(ns)),
                  "cm_seed_batch")
        k = self.n * 4 * ns.value
        return a[:k], b[:k], c[:k], ns.value

    def chains(self, slot):
        ch = np.zeros(self.n * 4 * CM_BESTCHAINLIM, dtype=CHAIN_DTYPE)
        nc = np.zeros(self.n * 4, np.int32)
        hh = np.zeros(self.n * 4, np.int32)
        self._chk(self.L.cm_chain_batch(self.h, slot, ch.ctypes.data, nc.ctypes.data, hh.ctypes.data), "cm_chain_batch")
        return ch, nc, hh

    def prof(self, on=True):
        self._chk(self.L.cm_prof_enable(self.h, int(on)), "cm_prof_enable")

    def prof_reset(self):
        self._chk(self.L.cm_prof_reset(self.h), "cm_prof_reset")

    def prof_get(self):
        ms = (C.c_double * 8)()
        n = (C.c_uint64 * 8)()
        self._chk(self.L.cm_prof_get(self.h, ms, n), "cm_prof_get")
        cnt = (C.c_uint64 * 8)()
        self._chk(self.L.cm_prof_counters(self.h, cnt), "cm_prof_counters")
        return list(ms), list(n), list(cnt)

    def close(self):
        if self.h:
            self._col_idx = self._col_st = self._rec = None
            self._rec_cap = 0
            self._col_cap = 0
            for p in self._pinned:
                self.L.cm_host_free(self.h, p)
            self._pinned = []
            self.L.cm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
