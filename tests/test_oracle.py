"""Oracle self-checks.  PARITY UNPINNED upstream (the reference ships no tests / fixtures and cannot be
built here), so these pin the oracle against planted truth of the synthetic data, brute-force
definitions of its DPs, and committed golden vectors of its own output (regression guard)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from circminer_amd import lib as cl
from oracle import oracle_py as op

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_seed21.json")


def _levenshtein_band(s, t, w):
    """Banded Levenshtein |i-j| <= w with unit costs, N never matching."""
    n, m = len(s), len(t)
    INF = 10 ** 7
    dp = [[INF] * (m + 1) for _ in range(n + 1)]
    for i in range(min(n, w) + 1):
        dp[i][0] = i
    for j in range(min(m, w) + 1):
        dp[0][j] = j
    for i in range(1, n + 1):
        for j in range(max(1, i - w), min(m, i + w) + 1):
            d = 0 if (s[i - 1] == t[j - 1] and chr(s[i - 1]).upper() in "ACGT") else 1
            dp[i][j] = min(dp[i - 1][j - 1] + d, dp[i - 1][j] + 1, dp[i][j - 1] + 1)
    return dp


def test_edit_dp_matches_bruteforce():
    O = op.load()
    rng = np.random.default_rng(3)
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    P = cl.default_params()
    for _ in range(300):
        m = int(rng.integers(8, 60))
        t = A[rng.integers(0, 4, m)]
        s = t.copy()
        for _k in range(int(rng.integers(0, 4))):
            p = int(rng.integers(0, len(s)))
            s[p] = A[rng.integers(0, 4)]
        if rng.random() < 0.3:
            s = np.delete(s, int(rng.integers(0, len(s))))
        s = np.concatenate([s, A[rng.integers(0, 4, 6)]])[:m + 3]
        n = len(s)
        if n <= 6 or n <= m:
            continue
        indel, sc = C.c_int(), C.c_int()
        ed = O.oracle_edit_side(C.byref(P), np.ascontiguousarray(s).ctypes.data, n, t.ctypes.data, m, 0, C.byref(indel), C.byref(sc))
        dp = _levenshtein_band(list(s), list(t), 3)
        cands = [(dp[i][m], m - i) for i in range(max(0, m - 3), min(m + 3, n) + 1) if dp[i][m] <= 4]
        if not cands:
            assert ed == 5 and indel.value == 4
        else:
            best = min(cands, key=lambda x: (2 * x[0], x[0], abs(x[1]), 0))
            bed = min(c[0] for c in cands)
            assert ed == bed
            # ties: smaller |indel| wins, and +1 beats -1 (first seen)
            ties = [c for c in cands if c[0] == bed]
            amin = min(abs(c[1]) for c in ties)
            want = [c[1] for c in ties if abs(c[1]) == amin][0]
            assert indel.value == want, (cands, indel.value)
            assert sc.value == -ed and best is not None


def test_one_side_is_hamming_when_w0():
    O = op.load()
    rng = np.random.default_rng(4)
    A = np.frombuffer(b"ACGT", dtype=np.uint8)
    for _ in range(100):
        n = int(rng.integers(1, 80))
        s = A[rng.integers(0, 4, n)]
        t = s.copy()
        t[rng.random(n) < 0.1] = ord("A")
        assert O.oracle_one_side(s.ctypes.data, n, t.ctypes.data, n, 0) == int((s != t).sum())


def test_planted_truth(ds_tiny):
    P = cl.default_params()
    st, act, cats = op.map_all_rounds(P, ds_tiny.hi, ds_tiny.batch)
    d = ds_tiny.d
    tx = d.src == 0
    conc = st["type"] == cl.CAT["CONCRD"]
    assert conc[tx].mean() > 0.93
    lo = np.minimum(st["spos_r1"], st["spos_r2"]).astype(np.int64)
    hi = np.maximum(st["epos_r1"], st["epos_r2"]).astype(np.int64)
    ok = (st["chr_id"] == d.truth_chr) & (np.abs(lo - d.truth_lo) <= 12) & (np.abs(hi - d.truth_hi) <= 12)
    assert ok[conc].mean() > 0.99                     # concordant pairs sit at the planted coordinates
    bs = d.src == 2
    assert np.isin(st["type"][bs], [cl.CAT["CHIBSJ"], cl.CAT["CHI2BSJ"], cl.CAT["CHIORF"]]).mean() > 0.85
    assert (act == np.isin(st["type"], [3, 4])).all()  # last round: only BSJ candidates are re-queued
    assert (st["ed_r1"][conc] + st["ed_r2"][conc] <= 8).all()


def test_round_carry_semantics(ds_tiny2r):
    """Two packed contigs -> two rounds; retired pairs are never touched again and unmapped types
    carry only `type` into the next round (fastq_parser.cpp:243-267)."""
    P = cl.default_params()
    b, hi = ds_tiny2r.batch, ds_tiny2r.hi
    st, act = op.default_state(P, b.n)
    c0 = op.map_round(P, hi.views[0], hi.annots[0], b, False, st, act)
    retired = act == 0
    assert (c0[retired] == 0).all() and retired.sum() > 0          # scan level 0: CONCRD pairs retire
    snap = st.copy()
    unm = (act == 1) & ~np.isin(st["type"], [0, 1, 2, 3, 4, 5, 7])
    assert (st["tlen"][unm] == 10 ** 9).all() and (st["chr_id"][unm] == -1).all()
    c1 = op.map_round(P, hi.views[1], hi.annots[1], b, True, st, act)
    assert (c1[retired] == -1).all() and st[retired].tobytes() == snap[retired].tobytes()
    assert (st["type"] <= snap["type"]).all()                       # a round can only improve the category


def test_golden_vectors(ds_tiny):
    """Regression guard: digest of the oracle's own output on the committed seed (tests/golden)."""
    P = cl.default_params()
    st, act, cats = op.map_all_rounds(P, ds_tiny.hi, ds_tiny.batch)
    ch, n, h = op.chains(P, ds_tiny.hi.views[0], ds_tiny.hi.annots[0], ds_tiny.batch)
    cur = {"n_pairs": int(ds_tiny.batch.n), "state_sha256": hashlib.sha256(st.tobytes()).hexdigest(),
           "type_hist": np.bincount(st["type"], minlength=14).tolist(), "nchain_sum": int(n.sum()),
           "chain_sha256": hashlib.sha256(ch.tobytes()).hexdigest()}
    if not os.path.exists(GOLD):
        pytest.skip("golden file missing: run tests/golden/make_golden.py")
    gold = json.load(open(GOLD))
    assert cur == gold
