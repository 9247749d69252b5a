"""Shared fixtures.  `-m "not gpu"` runs on the build box (no GPU): oracle, host builders, the
host emulation of the kernel bodies, C-ABI surface.  `-m gpu` are the parity tests proper: the
HIP path through the C-ABI against the oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")    # see bench.py: before torch brings up HIP
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from circminer_amd import _build, lib as cl, synth  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    _install_fault_harness()


def _install_fault_harness():
    """One occurrence of an abort must be enough to name its cause (round 3 lost one: the runtime's message went into pytest's
    captured stderr and died with the process).  tests/harness/fault_harness.c runs on the faulting thread and leaves
    gpurun_out/faults/fault_<pid>.txt: native stack of that thread + the captured stderr; Python's faulthandler adds the Python
    stacks of all threads to gpurun_out/faults/py_<pid>.txt."""
    import faulthandler
    out_dir = os.path.join(ROOT, "gpurun_out", "faults")
    try:
        os.makedirs(out_dir, exist_ok=True)
        so_dir = os.path.join(ROOT, "tests", "_hostemu")
        os.makedirs(so_dir, exist_ok=True)
        so = os.path.join(so_dir, "libfaultharness.so")
        src = os.path.join(ROOT, "tests", "harness", "fault_harness.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", "-rdynamic", src, "-o", so])
        H = C.CDLL(so)
        H.cm_fault_harness_install.argtypes = [C.c_char_p]
        if H.cm_fault_harness_install(out_dir.encode()) != 0:
            return
        global _fault_py_file
        _fault_py_file = open(os.path.join(out_dir, f"py_{os.getpid()}.txt"), "w")
        faulthandler.enable(file=_fault_py_file, all_threads=True)        # runs first, then hands the signal on to the C handler
    except Exception as e:                                                 # the harness must never be the reason a suite fails
        sys.stderr.write(f"[conftest] fault harness not installed: {e}\n")


def pytest_unconfigure(config):
    f = globals().get("_fault_py_file")
    if f is not None and f.tell() == 0:                                   # nothing happened: leave no empty files behind
        name = f.name
        f.close()
        try:
            os.remove(name)
        except OSError:
            pass


@pytest.fixture(scope="session", autouse=True)
def built():
    _build.build()
    from oracle import oracle_py
    oracle_py.build()
    return True


@pytest.fixture(scope="session")
def emu(built):
    return load_emu()


@pytest.fixture(scope="session")
def emu24(built):
    return load_emu(wide=True)


def load_emu(wide=False):
    """Host emulation of the kernel bodies (tests/hostemu.cpp) — test infrastructure only.  wide: the kernel bodies as the
    library's second build compiles them (reads of up to 24 seeds, -DCM_MAX_CHAIN_FRAGS=24)."""
    out_dir = os.path.join(ROOT, "tests", "_hostemu")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libcmemu24.so" if wide else "libcmemu.so")
    if os.environ.get("CM_EMU_LIB") and not wide:           # e.g. an ASan / UBSan build of tests/hostemu.cpp (see tests/diag/asan_emu.sh)
        so = os.environ["CM_EMU_LIB"]
    srcs = [os.path.join(ROOT, "tests", "hostemu.cpp"), os.path.join(ROOT, "circminer_amd", "csrc", "cm_core.h"),
            os.path.join(ROOT, "include", "circminer_hot.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "circminer_amd", "csrc"), srcs[0], "-o", so] + (["-DCM_MAX_CHAIN_FRAGS=24"] if wide else []))
    E = C.CDLL(so)
    vp, pp = C.c_void_p, C.POINTER
    E.emu_seed_batch.argtypes = [pp(cl.Params), pp(cl.IndexView), pp(cl.Reads), C.c_uint32, vp, vp, vp]
    E.emu_chain_batch.argtypes = [pp(cl.Params), pp(cl.IndexView), pp(cl.AnnotView), pp(cl.Reads), vp, vp, vp]
    E.emu_map_round.argtypes = [pp(cl.Params), pp(cl.IndexView), pp(cl.AnnotView), pp(cl.Reads), C.c_int, vp, vp, vp]
    E.emu_edit_side.argtypes = [pp(cl.Params), vp, C.c_int, vp, C.c_int, C.c_int, pp(C.c_int), pp(C.c_int)]
    E.emu_drop_sc.argtypes = [pp(cl.Params), vp, C.c_int, vp, C.c_int, C.c_int, pp(C.c_int), pp(C.c_int), pp(C.c_int)]
    E.emu_one_side.argtypes = [pp(cl.Params), vp, C.c_int, vp, C.c_int, C.c_int]
    E.emu_leftovers_matter.argtypes = [C.c_int] * 5
    E.emu_leftover_type.argtypes = [C.c_int] * 4
    return E


class DataSet:
    def __init__(self, tmpdir, preset, n_pairs, seed, kmer=20, **kw):
        self.d = synth.generate(preset, n_pairs=n_pairs, seed=seed, **kw)
        self.gtf = os.path.join(str(tmpdir), f"{preset}_{seed}.gtf")
        with open(self.gtf, "w") as f:
            f.write(self.d.gtf_text)
        self.hi = cl.HostIndex(self.d.contigs, self.d.chr_table, self.gtf, kmer=kmer)
        self.batch = cl.ReadBatch(self.d.seq1, self.d.seq2)
        self.kmer = kmer
        # What the ORACLE runs on: index + annotation from its own builders (oracle/cm_oracle_build.cpp, written from the reference's
        # HashTable.c / gene_annotation.cpp / interval_tree_impl.h), while the HIP path gets the product's (host_index.cpp /
        # host_annot.cpp).  Single-threaded: small genomes only; the big presets share the product's views.
        from oracle import oracle_py
        self.ohi = oracle_py.OracleIndex(self.d.contigs, self.d.chr_table, self.gtf, kmer=kmer) if sum(len(c) for c in self.d.contigs) <= 8_000_000 else self.hi


@pytest.fixture(scope="session")
def ds_tiny(tmp_path_factory, built):
    return DataSet(tmp_path_factory.mktemp("tiny"), "tiny", 1500, 21)


@pytest.fixture(scope="session")
def ds_tiny2r(tmp_path_factory, built):
    return DataSet(tmp_path_factory.mktemp("tiny2r"), "tiny2r", 1200, 22)


@pytest.fixture(scope="session")
def ds_variety(tmp_path_factory, built):
    """two packed contigs with nested / overlapping / opposite-strand / single-exon / duplicate-span genes, an exon next to the
    chromosome start and GTF gene blocks out of coordinate order (synth.add_variety)"""
    return DataSet(tmp_path_factory.mktemp("variety"), "variety", 4000, 31, mix=(0.5, 0.2, 0.3))


@pytest.fixture(scope="session")
def ds_long(tmp_path_factory, built):
    """2 x 300 bp reads with a k = 14 index: 21 seeds per read, the most the reference's command line allows (maxReadLength 300,
    k >= 14; src/commandline_parser.cpp:14,242-247) -- the library's 24-seed build"""
    return DataSet(tmp_path_factory.mktemp("long"), "tiny2r", 900, 41, kmer=14, read_len=300)


@pytest.fixture(scope="session")
def ds_small(tmp_path_factory, built):
    return DataSet(tmp_path_factory.mktemp("small"), "small", 20000, 23)


def states_equal(a, b):
    return a.tobytes() == b.tobytes()


def first_diff(a, b):
    for i in range(len(a)):
        if a[i].tobytes() != b[i].tobytes():
            return i, a[i], b[i]
    return None


class _Shim:
    """DataSet-like wrapper: same index / annotation, another read batch."""

    def __init__(self, ds, batch):
        self.d, self.hi, self.ohi, self.kmer, self.batch = ds.d, ds.hi, ds.ohi, ds.kmer, batch


@pytest.fixture(scope="session")
def ds_dirty(ds_tiny2r):
    """Edge cases of the read side on the two-contig data set: ragged lengths (0 .. 300 bp, i.e. no seed at all up to
    15 seeds), N runs, lower-case stretches (valid for the DPs, invalid for forward-strand seeds, valid again on the
    reverse strand after FASTQParser::set_comp), reads cut inside a seed, one mate much shorter than the other."""
    d = ds_tiny2r.d
    rng = np.random.default_rng(99)
    n = 600
    src = rng.integers(0, d.seq1.shape[0], n)
    s1, s2, l1, l2 = [], [], [], []

    def mangle(read, i):
        r = read.copy()
        k = i % 12
        if k == 0:
            r = r[:0]                                   # empty read
        elif k == 1:
            r = r[:int(rng.integers(1, 20))]            # shorter than one seed
        elif k == 2:
            r = r[:int(rng.integers(20, 40))]           # exactly one seed
        elif k == 3:
            r = r[:int(rng.integers(41, 150))]          # cut inside a seed
        elif k == 4:
            a = int(rng.integers(0, 120)); r[a:a + int(rng.integers(1, 25))] = ord("N")
        elif k == 5:
            a = int(rng.integers(0, 100)); r[a:a + 40] = np.frombuffer(bytes(r[a:a + 40]).lower(), np.uint8)
        elif k == 6:
            r = np.frombuffer(bytes(r).lower(), np.uint8).copy()
        elif k == 7:
            r = np.concatenate([r, r])                  # 300 bp: 15 seeds per orientation
        elif k == 8:
            r = np.concatenate([r, r[:int(rng.integers(1, 140))]])
        elif k == 9:
            r[int(rng.integers(0, 150))] = ord("n")
        return r

    for i in range(n):
        a, b = mangle(d.seq1[src[i]], i), mangle(d.seq2[src[i]], (i * 7 + 3) % 12 + 12 * (i % 5 == 0) * 0)
        s1.append(a); s2.append(b); l1.append(len(a)); l2.append(len(b))
    batch = cl.ReadBatch(np.concatenate(s1) if sum(l1) else np.zeros(0, np.uint8), np.concatenate(s2) if sum(l2) else np.zeros(0, np.uint8),
                         np.array(l1), np.array(l2))
    return _Shim(ds_tiny2r, batch)
