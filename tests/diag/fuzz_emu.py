"""Long-running parity fuzz on the CPU: the kernel bodies (host emulation, tests/hostemu.cpp) against the oracle over many
random genomes / annotations / read sets and parameter draws, all rounds.  Not part of the test suite (minutes to hours);
prints one line per data set and stops at the first difference.  With CM_FUZZ_GPU=1 the HIP path through the C ABI is the
one compared with the oracle (run on the GPU box).
usage: python tests/diag/fuzz_emu.py [first_seed] [n_sets] [pairs]"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
from circminer_amd import lib as cl  # noqa: E402
from oracle import oracle_py as op  # noqa: E402

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_sets = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
from circminer_amd import _build  # noqa: E402
_build.build()
op.build()
GPU = bool(os.environ.get("CM_FUZZ_GPU"))
E = None if GPU else conftest.load_emu()
td = tempfile.mkdtemp()
for k in range(n_sets):
    seed = s0 + k
    rng = np.random.default_rng(seed)
    preset = os.environ.get("CM_FUZZ_PRESET") or ["tiny", "tiny2r"][k % 2]
    mix = [(0.70, 0.25, 0.05), (0.3, 0.2, 0.5), (0.1, 0.8, 0.1), (0.5, 0.0, 0.5)][int(rng.integers(0, 4))]
    kw = [dict(), dict(scan_level=1), dict(scan_level=2, max_ed=6), dict(band=2), dict(band=5, max_ed=6), dict(max_sc=3, max_tlen=300),
          dict(max_chain_len=5), dict(seed_lim=50), dict(max_intron=20000)][int(rng.integers(0, 9))]
    read_len = int(rng.choice([76, 100, 125, 150, 151, 250]))
    t = time.time()
    ds = conftest.DataSet(td, preset, pairs, seed, mix=mix, read_len=read_len, fam_copies=int(rng.choice([6, 40, 400])))
    P = cl.default_params(**kw)
    dirty = int(rng.integers(0, 3)) == 0
    if dirty:            # ragged lengths, N runs, lower-case stretches, reads longer than the generator's (cf. conftest.ds_dirty)
        s1, s2 = [], []
        for arr, dst in ((ds.d.seq1, s1), (ds.d.seq2, s2)):
            for i in range(arr.shape[0]):
                r = arr[i].copy()
                k = int(rng.integers(0, 14))
                if k == 0:
                    r = r[:int(rng.integers(0, 45))]
                elif k == 1:
                    r = r[:int(rng.integers(45, len(r) + 1))]
                elif k == 2:
                    a = int(rng.integers(0, len(r))); r[a:a + int(rng.integers(1, 25))] = ord("N")
                elif k == 3:
                    a = int(rng.integers(0, len(r))); r[a:a + 40] = np.frombuffer(bytes(r[a:a + 40]).lower(), np.uint8)
                elif k == 4:
                    r = np.concatenate([r, r[:int(rng.integers(1, 300 - len(r) + 1))]]) if len(r) < 300 else r
                elif k == 5:
                    r[int(rng.integers(0, len(r)))] = ord("acgtn"[int(rng.integers(0, 5))])
                dst.append(r)
        l1, l2 = np.array([len(x) for x in s1]), np.array([len(x) for x in s2])
        ds.batch = cl.ReadBatch(np.concatenate(s1), np.concatenate(s2), l1, l2)
    st0, act0 = op.default_state(P, ds.batch.n)
    st1, act1 = st0.copy(), act0.copy()
    hp = None
    if GPU:
        hp = cl.HotPath(P)
        for ci in range(ds.hi.n_contigs):
            hp.load_contig(ci, ds.hi.views[ci], ds.hi.annots[ci])
        hp.upload(ds.batch)
    for ci in range(ds.hi.n_contigs):
        last = ci == ds.hi.n_contigs - 1
        iv, av = ds.hi.views[ci], ds.hi.annots[ci]
        cat0 = op.map_round(P, iv, av, ds.batch, last, st0, act0)
        if GPU:
            hp.map_round(ci, last)
            st1, cat1, act1 = hp.download()
        else:
            cat1 = np.full(ds.batch.n, -1, np.int32)
            rc = E.emu_map_round(C.byref(P), C.byref(iv), C.byref(av), C.byref(ds.batch.c), int(last), st1.ctypes.data, act1.ctypes.data, cat1.ctypes.data)
            assert rc == 0, rc
        if not ((cat0 == cat1).all() and (act0 == act1).all() and st0.tobytes() == st1.tobytes()):
            print("MISMATCH", seed, preset, mix, kw, read_len, "round", ci, conftest.first_diff(st0, st1), flush=True)
            sys.exit(1)
    if GPU:          # and all rounds in one call (pipelined; several tiles per batch are walked round-major: CM_TILE_PAIRS)
        hp.reset()
        hp.map_rounds(list(range(ds.hi.n_contigs)))
        st2, cat2, act2 = hp.download()
        if not ((cat0 == cat2).all() and (act0 == act2).all() and st0.tobytes() == st2.tobytes()):
            print("MISMATCH (map_rounds)", seed, preset, mix, kw, read_len, conftest.first_diff(st0, st2), flush=True)
            sys.exit(1)
    print(f"seed {seed} {preset}{' dirty' if dirty else ''} mix {mix} {kw} len {read_len}: ok, types {np.bincount(st0['type'], minlength=14).tolist()} ({time.time() - t:.0f}s)", flush=True)
    if hp is not None:
        hp.close()
    ds.hi.close()
print("all equal")
