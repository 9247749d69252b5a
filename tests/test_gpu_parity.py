"""GPU parity tests proper: HIP path through the C-ABI vs the CPU oracle (bit-exact)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from circminer_amd import lib as cl
from oracle import oracle_py as op
from conftest import first_diff

pytestmark = pytest.mark.gpu

# A process that uses both this library and PyTorch's device tensors (the BsjGather test below, bench.py) has to let
# torch bring up its bundled HIP runtime first: initialised second, it reports "No HIP GPUs are available".
if torch.cuda.is_available():
    torch.cuda.init()


def _chains_equal(c0, n0, c1, n1):
    assert (n0 == n1).all()
    a, b = c0.reshape(-1, cl.CM_BESTCHAINLIM), c1.reshape(-1, cl.CM_BESTCHAINLIM)
    for r in np.nonzero(n0)[0]:
        for k in range(n0[r]):
            x, y = a[r, k], b[r, k]
            L = int(x["chain_len"])
            assert L == int(y["chain_len"]), (r, k)
            assert x["score"] == y["score"], (r, k, x["score"], y["score"])      # fp32 of the fp64 sum, exact
            assert (x["rpos"][:L] == y["rpos"][:L]).all() and (x["qpos"][:L] == y["qpos"][:L]).all(), (r, k)


def _run_all_rounds(ds, P):
    hp = cl.HotPath(P)
    st0, act0 = op.default_state(P, ds.batch.n)
    hp.upload(ds.batch)
    for ci in range(ds.hi.n_contigs):
        last = ci == ds.hi.n_contigs - 1
        hp.load_contig(ci, ds.hi.views[ci], ds.hi.annots[ci])
        cat0 = op.map_round(P, ds.ohi.views[ci], ds.ohi.annots[ci], ds.batch, last, st0, act0)
        hp.map_round(ci, last)
        st1, cat1, act1 = hp.download()
        assert (cat0 == cat1).all(), np.nonzero(cat0 != cat1)[0][:10]
        assert (act0 == act1).all()
        assert st0.tobytes() == st1.tobytes(), first_diff(st0, st1)
        idx, stc = hp.collect_active()                 # stable device-side compaction of the re-queued pairs
        want = np.nonzero(act1)[0]
        assert (idx == want).all() and stc.tobytes() == st1[want].tobytes()
        rec = hp.collect_records(1000)                 # the same pairs as device-assembled records
        assert (rec["pair"] == want + 1000).all() and rec["state"].tobytes() == st1[want].tobytes()
        rec = rec.copy()
        # ... and left in caller-owned HBM for the gather to rank 0 (dist.BsjGather; single process here)
        from circminer_amd import dist as cdist
        g = cdist.BsjGather(ds.batch.n, torch.device("cuda", 0))
        for _ in range(2):                             # buffers are reused from batch to batch
            n = hp.collect_records_device(1000, ds.batch.n, g.send_ptr())
            g.submit(n)
            assert n == len(want) and g.result().tobytes() == rec.tobytes()
        host = np.zeros(4, dtype=cl.RECORD_DTYPE)     # a host pointer is refused, not dereferenced by a kernel
        with pytest.raises(RuntimeError):
            hp.collect_records_device(0, 4, host.ctypes.data)
    hp.close()
    return st0


@pytest.mark.parametrize("name", ["ds_tiny", "ds_small"])
def test_seed_parity(name, request):
    ds = request.getfixturevalue(name)
    P = cl.default_params(kmer=ds.kmer)
    hp = cl.HotPath(P)
    hp.load_contig(0, ds.hi.views[0], ds.hi.annots[0])
    hp.upload(ds.batch)
    a1, b1, c1, S = hp.seeds(0)
    a0, b0, c0 = op.seeds(P, ds.ohi.views[0], ds.batch, S)
    assert (c0 == c1).all() and (b0 == b1).all()
    m = c0 > 0
    assert (a0[m] == a1[m]).all()
    hp.close()


@pytest.mark.parametrize("name", ["ds_tiny", "ds_small"])
def test_chain_parity(name, request):
    ds = request.getfixturevalue(name)
    P = cl.default_params(kmer=ds.kmer)
    hp = cl.HotPath(P)
    hp.load_contig(0, ds.hi.views[0], ds.hi.annots[0])
    hp.upload(ds.batch)
    c1, n1, h1 = hp.chains(0)
    c0, n0, h0 = op.chains(P, ds.ohi.views[0], ds.ohi.annots[0], ds.batch)
    assert (h0 == h1).all()
    _chains_equal(c0, n0, c1, n1)
    hp.close()


@pytest.mark.parametrize("name", ["ds_tiny", "ds_tiny2r", "ds_small", "ds_variety"])
def test_map_parity_all_rounds(name, request):
    ds = request.getfixturevalue(name)
    st = _run_all_rounds(ds, cl.default_params(kmer=ds.kmer))
    # planted truth: most transcriptomic pairs come back concordant at the planted coordinates
    m = (ds.d.src == 0)
    assert (st["type"][m] == cl.CAT["CONCRD"]).mean() > (0.9 if name != "ds_variety" else 0.8)


@pytest.mark.parametrize("kw", [dict(scan_level=1), dict(scan_level=2, max_ed=8, seed_lim=1000), dict(max_chain_len=5, max_tlen=300)])
def test_map_parity_param_variants(ds_tiny2r, kw):
    _run_all_rounds(ds_tiny2r, cl.default_params(kmer=ds_tiny2r.kmer, **kw))


def test_map_batch_wrapper_and_errors(ds_tiny):
    P = cl.default_params()
    hp = cl.HotPath(P)
    with pytest.raises(RuntimeError):
        hp.upload(ds_tiny.batch)
        hp.map_round(3, True)           # slot not loaded -> CM_ESTATE, never a crash
    hp.load_contig(0, ds_tiny.hi.views[0], ds_tiny.hi.annots[0])
    st = np.zeros(ds_tiny.batch.n, dtype=cl.MAPPED_DTYPE)
    cat = np.zeros(ds_tiny.batch.n, dtype=np.int32)
    import ctypes as C
    rc = hp.L.cm_map_batch(hp.h, 0, 1, C.byref(ds_tiny.batch.c), None, st.ctypes.data, cat.ctypes.data)
    assert rc == 0
    st0, act0 = op.default_state(P, ds_tiny.batch.n)
    cat0 = op.map_round(P, ds_tiny.hi.views[0], ds_tiny.hi.annots[0], ds_tiny.batch, True, st0, act0)
    assert (cat0 == cat).all() and st0.tobytes() == st.tobytes()
    hp.close()


@pytest.mark.parametrize("n_pairs,seed,kw", [(1_000_000, 21, {}), (300_000, 77, dict(scan_level=1)), (200_000, 78, dict(scan_level=2, max_ed=6, seed_lim=1000))])
def test_full_size_parity_chr21(tmp_path_factory, n_pairs, seed, kw):
    """BASELINE.json configs[1] at full size: chr21-like contig, 1 M pairs — every pair's state, category and
    active flag bit-exact against the oracle (run on all host cores of the GPU box); two more genomes / read sets
    at other scan levels."""
    import os
    import threading
    from conftest import DataSet
    ds = DataSet(tmp_path_factory.mktemp("chr21"), "chr21", n_pairs, seed)
    if not kw:
        # At this size too the oracle runs on index + annotation from ITS OWN builders (oracle/cm_oracle_build.cpp, single thread,
        # from src/mrsfast/HashTable.c:769-839 + Sort.c:116-117 and gene_annotation.cpp / interval_tree_impl.h), compared array by
        # array with the product's (the range-partitioned parallel build of host_index.cpp) before anything is mapped.
        import time
        from builders_util import assert_host_views_equal
        t = time.time()
        ds.ohi = op.OracleIndex(ds.d.contigs, ds.d.chr_table, ds.gtf, kmer=ds.kmer)
        assert_host_views_equal(ds.hi, ds.ohi)
        print(f"oracle's own builders on {len(ds.d.contigs[0])} bp: {time.time() - t:.0f}s, views identical to the product's", flush=True)
    P = cl.default_params(**kw)
    hp = cl.HotPath(P)
    hp.load_contig(0, ds.hi.views[0], ds.hi.annots[0])
    hp.upload(ds.batch)
    hp.map_round(0, True)
    st1, cat1, act1 = hp.download()
    n = ds.batch.n
    st0, act0 = op.default_state(P, n)
    cat0 = np.full(n, -1, np.int32)
    T = max(1, os.cpu_count() or 1)
    L = op.load()
    import ctypes as C

    def work(a, b):
        L.oracle_map_round(C.byref(P), C.byref(ds.ohi.views[0]), C.byref(ds.ohi.annots[0]), C.byref(ds.batch.c), 1, st0.ctypes.data,
                           act0.ctypes.data, cat0.ctypes.data, a, b)

    th = [threading.Thread(target=work, args=((n * i) // T, (n * (i + 1)) // T)) for i in range(T)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert (cat0 == cat1).all() and (act0 == act1).all()
    assert st0.tobytes() == st1.tobytes(), first_diff(st0, st1)
    # size-independent properties at full size
    assert (act1 == np.isin(st1["type"], [3, 4])).all()
    idx, stc = hp.collect_active()
    assert (idx == np.nonzero(act1)[0]).all()
    hp.close()


def test_batches_beyond_two_tiles_worth_of_pairs(tmp_path_factory, monkeypatch):
    """A batch of more than 2^21 pairs is walked in two tiles of half the batch (up to 2^21 pairs each; cm_hot.hip tile_for); with
    CM_TILE_PAIRS=2^20 the same batch takes three tiles, whose seeds are computed a chain stage early (the flags they read are two
    items old).  Both walks must give the same bytes, equal the oracle on ranges around every tile boundary, and keep the
    size-independent properties."""
    import os
    import threading
    import ctypes as C
    from conftest import DataSet
    n = (1 << 21) + 200_000
    ds = DataSet(tmp_path_factory.mktemp("chr21big"), "chr21", n, 91)
    P = cl.default_params()

    def run():
        hp = cl.HotPath(P)
        hp.load_contig(0, ds.hi.views[0], ds.hi.annots[0])
        hp.upload(ds.batch)
        hp.map_rounds([0], True)
        st, cat, act = hp.download()
        idx = np.array(hp.collect_active()[0], copy=True)       # (the arrays collect_active returns live in the context)
        st, cat, act = st.copy(), cat.copy(), act.copy()
        hp.close()
        return st, cat, act, idx

    st_a, cat_a, act_a, idx_a = run()                       # two tiles of 1 179 648 pairs
    monkeypatch.setenv("CM_TILE_PAIRS", str(1 << 20))
    st_b, cat_b, act_b, idx_b = run()                       # three tiles
    assert st_a.tobytes() == st_b.tobytes() and (cat_a == cat_b).all() and (act_a == act_b).all()
    assert (act_a == np.isin(st_a["type"], [3, 4])).all()
    assert (idx_a == np.nonzero(act_a)[0]).all() and (idx_a == idx_b).all()
    # the oracle on 40 000 pairs around each tile boundary of either walk, and at both ends
    L = op.load()
    st0, act0 = op.default_state(P, n)
    cat0 = np.full(n, -1, np.int32)
    spans = [(0, 20_000), (1_179_648 - 20_000, 1_179_648 + 20_000), ((1 << 20) - 20_000, (1 << 20) + 20_000), ((1 << 21) - 20_000, (1 << 21) + 20_000),
             (n - 20_000, n)]
    T = max(1, min(64, os.cpu_count() or 1))
    jobs = []
    for a, b in spans:
        step = (b - a + T - 1) // T
        jobs += [(x, min(b, x + step)) for x in range(a, b, step)]

    def work(a, b):
        L.oracle_map_round(C.byref(P), C.byref(ds.hi.views[0]), C.byref(ds.hi.annots[0]), C.byref(ds.batch.c), 1, st0.ctypes.data,
                           act0.ctypes.data, cat0.ctypes.data, a, b)

    th = [threading.Thread(target=work, args=j) for j in jobs]
    [t.start() for t in th]
    [t.join() for t in th]
    for a, b in spans:
        assert (cat0[a:b] == cat_a[a:b]).all() and (act0[a:b] == act_a[a:b]).all()
        assert st0[a:b].tobytes() == st_a[a:b].tobytes(), first_diff(st0[a:b], st_a[a:b])


def test_config5_stress_params(tmp_path_factory):
    """BASELINE.json configs[4] flags on a small genome: k=22 --seed-lim 1000 --max-ed 8 --scan-lev 2."""
    from conftest import DataSet
    ds = DataSet(tmp_path_factory.mktemp("k22s"), "small", 8000, 37, kmer=22)
    _run_all_rounds(ds, cl.default_params(kmer=22, seed_lim=1000, max_ed=8, scan_level=2))


def test_mapping_from_index_files_matches_in_memory_index(ds_tiny2r, tmp_path):
    """contigs loaded from a stock-format index file (full and compact) map exactly like the
    in-memory builder's views (two packed contigs = two rounds)."""
    ds = ds_tiny2r
    packed = str(tmp_path / "ref.fa.packed.fa")
    with open(packed, "w") as f:
        for i, c in enumerate(ds.d.contigs):
            f.write(f">{i + 1}\n{c.tobytes().decode()}\n")
    P = cl.default_params(kmer=ds.kmer)

    def run(views_iter):
        hp = cl.HotPath(P)
        hp.upload(ds.batch)
        n = 0
        for ci, iv in enumerate(views_iter):
            assert iv.contig_num == ci
            hp.load_contig(ci, iv, ds.hi.annots[ci])
            hp.map_round(ci, ci == ds.hi.n_contigs - 1)
            hp.sync()
            n += 1
        assert n == ds.hi.n_contigs
        out = hp.download()
        hp.close()
        return out

    ref = run(iter(ds.hi.views))
    for compact in (False, True):
        idx = cl.write_index(packed, kmer=ds.kmer, compact=compact, n_threads=4)
        f = cl.IndexFile(idx, n_threads=4)
        assert f.kmer == ds.kmer and f.full == (not compact) and f.n_records == ds.hi.n_contigs
        got = run(f)
        f.close()
        assert got[0].tobytes() == ref[0].tobytes() and (got[1] == ref[1]).all() and (got[2] == ref[2]).all()


def test_index_table_flattened_on_the_device(ds_small, ds_tiny2r, tmp_path):
    """cm_host_next_contig_raw + cm_load_contig_raw: the table of a full-format index file goes over PCIe as it is in the file and
    the DEVICE builds bucket offsets (two scans) and the (checksum, position) arrays (a scatter at HBM bandwidth) -- what host
    threads do in cm_host_next_contig.  The resident index must be the same: every seed range of every probe (first entry, count,
    raw count: they pin bucket_off / checksum / pos) and all mapping results equal those of the host-flattened load.  A compact
    index file has no table: CM_EINVAL (the caller takes cm_host_next_contig); a corrupted header slot is refused."""
    for ds in (ds_small, ds_tiny2r):
        packed = str(tmp_path / f"ref{ds.hi.n_contigs}.fa.packed.fa")
        with open(packed, "w") as f:
            for i, c in enumerate(ds.d.contigs):
                f.write(f">{i + 1}\n{c.tobytes().decode()}\n")
        idx = cl.write_index(packed, kmer=ds.kmer, n_threads=4)
        P = cl.default_params(kmer=ds.kmer)
        res = {}
        for raw in (False, True):
            hp = cl.HotPath(P)
            hp.upload(ds.batch)
            f = cl.IndexFile(idx, n_threads=3, raw=raw)
            seeds = []
            for ci, rec in enumerate(f):
                assert rec.contig_num == ci
                if raw:
                    assert rec.n_buckets > 1000 and rec.table_slots > rec.n_buckets
                    hp.load_contig_raw(ci, rec, ds.hi.annots[ci])
                else:
                    hp.load_contig(ci, rec, ds.hi.annots[ci])
                seeds.append([x.copy() if hasattr(x, "copy") else x for x in hp.seeds(ci)])
                hp.map_round(ci, ci == ds.hi.n_contigs - 1)
                hp.sync()
            f.close()
            res[raw] = (seeds, hp.download())
            hp.close()
        for a, b in zip(res[False][0], res[True][0]):
            assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3] == b[3]
        assert res[False][1][0].tobytes() == res[True][1][0].tobytes() and (res[False][1][2] == res[True][1][2]).all()
    # compact format: no table in the file
    import shutil
    packed2 = packed + ".compact.fa"                   # (write_index names the index after the packed FASTA)
    shutil.copy(packed, packed2)
    idx2 = cl.write_index(packed2, kmer=ds.kmer, compact=True, n_threads=4)
    f = cl.IndexFile(idx2, n_threads=2, raw=True)
    with pytest.raises(RuntimeError):
        next(f)
    f.close()
    # a header slot that claims more entries than its bucket has slots
    f = cl.IndexFile(idx, n_threads=2, raw=True)
    rec = next(f)
    tab = np.ctypeslib.as_array(C.cast(rec.table, C.POINTER(C.c_int32)), (int(rec.table_slots) * 2,))
    tab[1] = 1 << 20                                   # info of the first bucket's header
    hp = cl.HotPath(P)
    with pytest.raises(RuntimeError, match="malformed"):
        hp.load_contig_raw(0, rec)
    hp.close()
    f.close()


def test_two_rounds_through_remain_fastq_files(ds_tiny2r, tmp_path):
    """the reference's way of carrying pairs between rounds (remain FASTQ + 23-token header,
    src/filter.cpp:413-455 -> src/fastq_parser.cpp:200-269) gives the same final states as the resident batch."""
    ds = ds_tiny2r
    P = cl.default_params(kmer=ds.kmer)
    n = ds.batch.n
    chrs = ds.d.chr_table
    # the input as FASTQ files
    p1, p2 = str(tmp_path / "in_1.fq"), str(tmp_path / "in_2.fq")
    for path, arr, mate in ((p1, ds.d.seq1, 1), (p2, ds.d.seq2, 2)):
        with open(path, "w") as f:
            for i in range(n):
                s = arr[i].tobytes().decode()
                f.write(f"@pair{i}/{mate}\n{s}\n+\n{'I' * len(s)}\n")
    # A: resident batch, both rounds
    hp = cl.HotPath(P)
    for ci in range(2):
        hp.load_contig(ci, ds.hi.views[ci], ds.hi.annots[ci])
    hp.upload(ds.batch)
    hp.map_round(0, False)
    hp.map_round(1, True)
    stA, catA, actA = hp.download()
    # B: round 1 from the FASTQ files, survivors through remain files, round 2 from those
    rd = cl.FastqReader(p1, p2, chrs, P.max_ed)
    b = rd.next_batch(n + 10)
    assert b.n == n and b.prior is None
    hp.upload(b)
    hp.map_round(0, False)
    st1, cat1, act1 = hp.download()
    idx, stc = hp.collect_active()
    idx = idx.copy()
    r1, r2 = str(tmp_path / "o_1_remain_R1.fastq"), str(tmp_path / "o_1_remain_R2.fastq")
    w = cl.RecordWriter(r1, r2, chrs)
    w.write_remain(b, st1, idx)
    w.close()
    pam = str(tmp_path / "o.mapping.pam")
    wp = cl.RecordWriter(pam, None, chrs)
    wp.write_pam(b, st1, np.nonzero(act1 == 0)[0])            # pairs retired by round 1 (skip)
    rd.close()
    rd2 = cl.FastqReader(r1, r2, chrs, P.max_ed)
    b2 = rd2.next_batch(n + 10)
    assert b2.n == len(idx) and b2.prior is not None and b2.prior.tobytes() == st1[idx.astype(np.int64)].tobytes()
    hp.upload(b2, b2.prior)
    hp.map_round(1, True)
    st2, cat2, act2 = hp.download()
    wp.write_pam(b2, st2)                                      # last round: every remaining pair is printed
    wp.close()
    rd2.close()
    hp.close()
    ii = idx.astype(np.int64)
    assert st2.tobytes() == stA[ii].tobytes() and (cat2 == catA[ii]).all() and (act2 == actA[ii]).all()
    done = np.nonzero(act1 == 0)[0]
    assert stA[done].tobytes() == st1[done].tobytes()
    rows = open(pam).read().strip().split("\n")
    assert len(rows) == n and sorted(r.split("\t")[0] for r in rows) == sorted(f"pair{i}" for i in range(n))


def test_multi_tile_batches_match_the_oracle(ds_small, monkeypatch):
    """Several launch groups per round (CM_TILE_PAIRS): the per-tile workspaces, the task pipeline of the mid
    pairs and the second stream are reused tile after tile; results must not depend on the tiling."""
    monkeypatch.setenv("CM_TILE_PAIRS", "4096")
    P = cl.default_params(kmer=ds_small.kmer)
    _run_all_rounds(ds_small, P)


def test_ragged_and_dirty_reads(ds_dirty):
    """empty / sub-seed / ragged / 300-bp reads, N runs and lower-case stretches (see conftest.ds_dirty)"""
    _run_all_rounds(ds_dirty, cl.default_params(kmer=ds_dirty.kmer))
    _run_all_rounds(ds_dirty, cl.default_params(kmer=ds_dirty.kmer, scan_level=2, max_ed=6))


def test_empty_batch_is_a_no_op(ds_tiny):
    hp = cl.HotPath(cl.default_params())
    hp.load_contig(0, ds_tiny.hi.views[0], ds_tiny.hi.annots[0])
    b = cl.ReadBatch(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.int64), np.zeros(0, np.int64))
    hp.upload(b)
    hp.map_round(0, True)
    st, cat, act = hp.download()
    idx, stc = hp.collect_active()
    assert len(st) == len(cat) == len(act) == len(idx) == 0
    hp.close()


def test_one_context_many_batches(ds_tiny):
    """A streaming caller re-uses one context for batch after batch (different sizes, an empty one in between): every
    per-batch buffer is re-made, the grow-only output staging survives, and each batch's results equal a fresh
    context's.  (Regression: the record staging buffer was freed with the batch while its capacity was kept.)"""
    d = ds_tiny.d
    P = cl.default_params()
    hp = cl.HotPath(P)
    hp.load_contig(0, ds_tiny.hi.views[0], ds_tiny.hi.annots[0])
    n = ds_tiny.batch.n
    want = {}
    for lo, hi_ in ((0, n // 2), (n // 2, n), (0, 0), (3, n // 4), (5, 6), (0, 63), (1, 65), (100, 165), (0, n)):   # 1 pair; around a wave
        b = cl.ReadBatch(d.seq1[lo:hi_], d.seq2[lo:hi_]) if hi_ > lo else \
            cl.ReadBatch(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(0, np.int64), np.zeros(0, np.int64))
        hp.upload(b)
        hp.map_round(0, True)
        st, cat, act = hp.download()
        rec = hp.collect_records(lo).copy()
        idx, stc = hp.collect_active()
        sel = np.nonzero(act)[0]
        assert (rec["pair"] == sel + lo).all() and rec["state"].tobytes() == st[sel].tobytes()
        assert (idx == sel).all() and stc.tobytes() == st[sel].tobytes()
        want[(lo, hi_)] = (st.copy(), cat.copy())
    hp.close()
    st_all, cat_all = want[(0, n)]
    for (lo, hi_), (st, cat) in want.items():       # process_read is a pure function of the pair: slices agree with the whole
        assert st.tobytes() == st_all[lo:hi_].tobytes() and (cat == cat_all[lo:hi_]).all()
    st0, act0 = op.default_state(P, n)
    op.map_round(P, ds_tiny.hi.views[0], ds_tiny.hi.annots[0], ds_tiny.batch, True, st0, act0)
    assert st0.tobytes() == st_all.tobytes()


def test_contigs_streamed_through_one_slot(ds_tiny2r):
    """The reference keeps one packed contig resident at a time (loadHashTable per round, src/circminer.cpp:258-268).
    Re-loading slot 0 for every round must give what the all-resident layout gives; the batch goes through twice so the
    second pass starts from buffers the first one left behind."""
    ds, P = ds_tiny2r, cl.default_params()
    want, _, _ = op.map_all_rounds(P, ds.ohi, ds.batch)
    hp = cl.HotPath(P)
    for _ in range(2):
        hp.upload(ds.batch)
        for ci in range(ds.hi.n_contigs):
            hp.load_contig(0, ds.hi.views[ci], ds.hi.annots[ci])
            hp.map_round(0, ci == ds.hi.n_contigs - 1)
        st, _, _ = hp.download()
        assert st.tobytes() == want.tobytes(), first_diff(want, st)
    hp.close()


@pytest.mark.parametrize("preset,contig_size,report", [("tiny2r", 150_000, 1), ("tiny", 1_100_000_000, 2)])
def test_stage1_from_files_to_files(preset, contig_size, report, tmp_path):
    """cm_mapping_run = the reference's mapping() (src/circminer.cpp:98-352) on stock file formats: FASTA -> packed genome +
    index file, GTF, gzip FASTQ in; <out>.mapping.pam / .sam and the last round's remain FASTQ out.  Expected rows come
    from the CPU oracle's final states through the Python restatements of the writers (tests/test_fastq_io.py)."""
    import gzip
    from circminer_amd import synth
    from test_fastq_io import py_pam_row, py_remain_header, py_sam_rows
    n = 3000
    d = synth.generate(preset, n_pairs=n, seed=33)
    fa = str(tmp_path / "ref.fa")
    with open(fa, "w") as f:
        for name, con, start, ln in d.chr_table:
            seq = d.contigs[con - 1][start:start + ln].tobytes().decode()
            f.write(f">{name} some description\n")
            f.writelines(seq[i:i + 70] + "\n" for i in range(0, ln, 70))
    packed, info = cl.pack_genome(fa, contig_size)
    idx = cl.write_index(packed, kmer=20, n_threads=4)
    gtf = str(tmp_path / "ref.gtf")
    open(gtf, "w").write(d.gtf_text)
    names = [f"frag.{i}" for i in range(n)]
    quals = ["".join(chr(33 + (i * 7 + k) % 40) for k in range(d.seq1.shape[1])) for i in range(16)]
    fq = []
    for mate, arr in ((1, d.seq1), (2, d.seq2)):
        p = str(tmp_path / f"reads_{mate}.fq.gz")
        with gzip.open(p, "wt", compresslevel=1) as f:
            for i in range(n):
                f.write(f"@{names[i]}/{mate} extra\n{arr[i].tobytes().decode()}\n+\n{quals[i % 16]}\n")
        fq.append(p)
    out = str(tmp_path / "run")
    P = cl.default_params(kmer=0)                                       # k comes from the index file
    st = cl.run_mapping(idx, gtf, fq[0], fq[1], out, P, report=report, n_threads=4, batch_pairs=1024)   # 3 batches

    # expectation: the oracle on the in-memory builder's views of the same genome
    gtf2 = str(tmp_path / "ref2.gtf")
    open(gtf2, "w").write(d.gtf_text)
    hi = op.OracleIndex(d.contigs, d.chr_table, gtf2, kmer=20)          # the oracle's own builders; the run above used the product's
    P20 = cl.default_params()
    want, act, _ = op.map_all_rounds(P20, hi, cl.ReadBatch(d.seq1, d.seq2))
    chrs = d.chr_table
    assert st.pairs == n and st.rounds == hi.n_contigs and st.bsj_pairs == int(act.sum()) > 0
    assert list(st.by_type) == [int((want["type"] == t).sum()) for t in range(14)]
    s1 = [d.seq1[i].tobytes().decode() for i in range(n)]
    s2 = [d.seq2[i].tobytes().decode() for i in range(n)]
    if report == 1:
        rows = open(out + ".mapping.pam").read().split("\n")
        assert rows[-1] == "" and rows[:-1] == [py_pam_row(names[i], want[i], chrs) for i in range(n)]
    else:
        rows = open(out + ".mapping.sam").read().split("\n")
        hdr = 1 + len(chrs)
        assert rows[0].startswith("@HD") and [r.split("\t")[1] for r in rows[1:hdr]] == [f"SN:{c[0]}" for c in chrs]
        exp = []
        for i in range(n):
            exp += py_sam_rows(names[i], want[i], chrs, s1[i], quals[i % 16], s2[i], quals[i % 16])
        assert rows[hdr:-1] == exp
    keep = np.nonzero(act)[0]
    assert set(want["type"][keep]) <= {3, 4}
    for mate, seqs in ((1, s1), (2, s2)):
        lines = open(f"{out}_{hi.n_contigs}_remain_R{mate}.fastq").read().split("\n")
        assert lines[-1] == "" and len(lines) == 4 * len(keep) + 1
        for k, i in enumerate(keep):
            assert lines[4 * k] == py_remain_header(names[i], want[i], chrs)
            assert lines[4 * k + 1:4 * k + 4] == [seqs[i], "+", quals[i % 16]]
    # stage 2 on the files the device path just wrote (cm_circ_run = circ_detect): candidates.pam and circ_report equal the
    # oracle's restatement of ProcessCirc fed with the GNU-sorted remain files
    from stage2_util import gnu_sort, oracle_stage2
    cs = cl.run_circ(idx, gtf, out, hi.n_contigs, cl.default_params(kmer=0))
    rem = [f"{out}_{hi.n_contigs}_remain_R{m}.fastq" for m in (1, 2)]
    want_c, want_r = oracle_stage2(tmp_path, hi, d, P20, gnu_sort(rem[0]), gnu_sort(rem[1]))
    assert open(out + ".candidates.pam", "rb").read() == want_c and open(out + ".circ_report", "rb").read() == want_r
    assert cs.pairs == len(keep) and cs.calls > 0 and want_r.count(b"\n") > 0
    # the same run from plain C++ (examples/cm_map.cpp: no Python, no torch in the process): identical files
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cm_map")
    libdir = os.path.join(root, "circminer_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "cm_map.cpp"), "-L", libdir,
                           "-lcmhot", f"-Wl,-rpath,{libdir}", "-o", exe])
    out2 = str(tmp_path / "run_cpp")
    msg = subprocess.check_output([exe, idx, gtf, fq[0], fq[1], out2, "pam" if report == 1 else "sam"], text=True)
    assert msg.startswith(f"{n} pairs, {hi.n_contigs} round(s), {int(act.sum())} BSJ")
    assert "stage 2:" in msg
    for suffix in ([".mapping.pam"] if report == 1 else [".mapping.sam"]) + [f"_{hi.n_contigs}_remain_R{m}.fastq" for m in (1, 2)] + \
            [".candidates.pam", ".circ_report"]:
        assert open(out + suffix, "rb").read() == open(out2 + suffix, "rb").read(), suffix
    # bad input is an error message, not a crash
    with pytest.raises(RuntimeError, match="k = 20"):
        cl.run_mapping(idx, gtf, fq[0], fq[1], out, cl.default_params(kmer=18))
    with pytest.raises(RuntimeError):
        cl.run_mapping(idx + ".nope", gtf, fq[0], fq[1], out, P)


def test_two_ranks_on_one_card_end_in_one_circ_report(tmp_path):
    """SURVEY 8(e) on the device path: two rank PROCESSES (both on cuda:0 here; one per GPU on a node) run cm_mapping_run with
    rank / world = their block of the plain-text FASTQ (cm_fastq_open_shard), each writing .part<rank> files; cm_merge_parts and
    one cm_circ_run on "rank 0".  Mapping file, remain files, candidates.pam and circ_report are the bytes of the one-process
    run, report 0 (records of the re-queued pairs only leave the device) and report 1 (PAM) alike."""
    import subprocess
    import sys
    from circminer_amd import synth
    from stage2_util import write_fastq_pair
    n = 6000
    d = synth.generate("tiny2r", n_pairs=n, seed=44, mix=(0.5, 0.2, 0.3))
    fa = str(tmp_path / "ref.fa")
    with open(fa, "w") as f:
        for name, con, start, ln in d.chr_table:
            f.write(f">{name}\n{d.contigs[con - 1][start:start + ln].tobytes().decode()}\n")
    packed, info = cl.pack_genome(fa, 150_000)
    idx = cl.write_index(packed, kmer=20, n_threads=4)
    gtf = str(tmp_path / "ref.gtf")
    open(gtf, "w").write(d.gtf_text)
    fq1, fq2 = write_fastq_pair(tmp_path, d, n)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from circminer_amd import lib as cl; "
            "st = cl.run_mapping(%r, %r, %r, %r, sys.argv[1], cl.default_params(kmer=0), report=int(sys.argv[4]), n_threads=4, batch_pairs=700, "
            "rank=int(sys.argv[2]), world=int(sys.argv[3])); print(st.pairs, st.bsj_pairs, list(st.by_type))") % (root, idx, gtf, fq1, fq2)
    files = {}
    for report in (0, 1):
        for world in (1, 2):
            out = str(tmp_path / f"r{report}w{world}")
            procs = [subprocess.Popen([sys.executable, "-c", code, out, str(r), str(world), str(report)], stdout=subprocess.PIPE, text=True)
                     for r in range(world)]
            outs = [p.communicate()[0] for p in procs]
            assert all(p.returncode == 0 for p in procs), outs
            stats = [o.strip().split(" ", 2) for o in outs]
            assert sum(int(x[0]) for x in stats) == n
            cl.merge_parts(out, 2, world, report)
            cs = cl.run_circ(idx, gtf, out, 2, cl.default_params(kmer=0))
            assert cs.pairs == sum(int(x[1]) for x in stats) > 100
            names = ["_2_remain_R1.fastq", "_2_remain_R2.fastq", ".candidates.pam", ".circ_report"] + ([".mapping.pam"] if report else [])
            files[(report, world)] = {s: open(out + s, "rb").read() for s in names}
            assert not [f for f in os.listdir(str(tmp_path)) if ".part" in f]
        assert files[(report, 1)] == files[(report, 2)]
    for s in ("_2_remain_R1.fastq", ".circ_report"):
        assert files[(0, 1)][s] == files[(1, 1)][s] and len(files[(0, 1)][s]) > 1000


def test_genes_with_more_than_64_isoforms(ds_tiny, tmp_path):
    """|common_tid| > 64 (cmc::TidList's overflow walk) on the device, light and heavy pair kernels alike."""
    from test_hostemu_parity import _many_isoforms
    sh = _many_isoforms(ds_tiny, tmp_path, 45)
    _run_all_rounds(sh, cl.default_params())


def test_staged_batches_overlap_and_match(ds_tiny2r, ds_dirty):
    """cm_reads_stage / cm_reads_swap: the next batch is copied on the copy stream while the resident one maps; every
    batch's results equal the oracle's, whatever was resident or staged before (ragged batch, carried prior states,
    a staged batch that is replaced before it is swapped in)."""
    ds = ds_tiny2r
    P = cl.default_params(kmer=ds.kmer)
    hp = cl.HotPath(P)
    for ci in range(ds.hi.n_contigs):
        hp.load_contig(ci, ds.hi.views[ci], ds.hi.annots[ci])
    n = ds.batch.n
    pa = hp.pinned_batch(ds.d.seq1[:n // 2], ds.d.seq2[:n // 2])
    pb = hp.pinned_batch(ds.d.seq1[n // 2:], ds.d.seq2[n // 2:])
    want, _, _ = op.map_all_rounds(P, ds.ohi, ds.batch)
    want_dirty, _, _ = op.map_all_rounds(P, ds.ohi, ds_dirty.batch)

    def rounds():
        for ci in range(ds.hi.n_contigs):
            hp.map_round(ci, ci == ds.hi.n_contigs - 1)

    with pytest.raises(RuntimeError):
        hp.swap()                                   # nothing staged
    hp.stage(pa)
    hp.swap()
    hp.stage(pb)                                    # in flight while A maps
    rounds()
    stA = hp.download()[0]
    hp.swap()
    hp.stage(ds_dirty.batch)                        # pageable, ragged
    rounds()
    stB = hp.download()[0]
    hp.swap()
    hp.stage(pb)                                    # replaced before it is used
    hp.stage(pa)
    rounds()
    stD = hp.download()[0]
    hp.swap()
    rounds()
    stA2 = hp.download()[0]
    assert stA.tobytes() == want[:n // 2].tobytes() and stA2.tobytes() == stA.tobytes()
    assert stB.tobytes() == want[n // 2:].tobytes()
    assert stD.tobytes() == want_dirty.tobytes()
    # carried states: round 1 resident, survivors staged with their prior state, round 2
    hp.upload(ds.batch)
    hp.map_round(0, False)
    st1, _, act1 = hp.download()
    idx = np.nonzero(act1)[0]
    sub = cl.ReadBatch(ds.d.seq1[idx], ds.d.seq2[idx])
    hp.stage(sub, np.ascontiguousarray(st1[idx]))
    hp.swap()
    hp.map_round(1, True)
    st2 = hp.download()[0]
    assert st2.tobytes() == want[idx].tobytes()
    hp.close()


@pytest.mark.parametrize("pool_max", [None, 1 << 18])
def test_improvement_log_pool_recovers(ds_small, monkeypatch, pool_max):
    """The chain improvement log has no capacity limit in the reference (score2chain, src/chain.cpp:191-206).  Here it comes out
    of a pool; when that runs out the stage is redone with a larger pool, and beyond CM_POOL_MAX in halves that get the whole
    pool each -- never a failed batch.  Forced with a 64 KB pool: results must still equal the oracle's."""
    monkeypatch.setenv("CM_POOL_BYTES", "65536")
    if pool_max:
        monkeypatch.setenv("CM_POOL_MAX", str(pool_max))
    _run_all_rounds(ds_small, cl.default_params(kmer=ds_small.kmer))


@pytest.mark.parametrize("tile", [None, "512"])
def test_rounds_in_one_call_match_round_by_round(ds_tiny2r, ds_small, ds_dirty, ds_variety, monkeypatch, tile):
    """cm_map_rounds: round r + 1 is seeded and chained (other streams, second set of chain buffers, flags of the round before)
    while the pair stage of round r runs.  Same final state, flags, categories and BSJ records as round-by-round calls and as
    the oracle; repeated on one context, with several tiles per batch, with a contig used twice and with an even / odd number
    of rounds (the active-flag arrays swap roles every round)."""
    if tile:
        monkeypatch.setenv("CM_TILE_PAIRS", tile)
    for ds, order in ((ds_tiny2r, [0, 1]), (ds_dirty, [0, 1]), (ds_tiny2r, [1, 0, 1]), (ds_small, [0]), (ds_tiny2r, [0, 1, 0, 1]), (ds_variety, [0, 1])):
        P = cl.default_params(kmer=ds.kmer)
        hp = cl.HotPath(P)
        for ci in range(ds.hi.n_contigs):
            hp.load_contig(ci, ds.hi.views[ci], ds.hi.annots[ci])
        st0, act0 = op.default_state(P, ds.batch.n)
        for k, ci in enumerate(order):
            cat0 = op.map_round(P, ds.ohi.views[ci], ds.ohi.annots[ci], ds.batch, k == len(order) - 1, st0, act0)
        for rep in range(2):
            hp.upload(ds.batch)
            hp.map_rounds(order, True)
            st1, cat1, act1 = hp.download()
            assert st0.tobytes() == st1.tobytes(), first_diff(st0, st1)
            assert (act0 == act1).all() and (cat0 == cat1).all()
            rec = hp.collect_records(7)
            keep = np.nonzero(act1)[0]
            assert (rec["pair"] == keep + 7).all() and rec["state"].tobytes() == st1[keep].tobytes()
        # and interleaved with single-round calls on the same context
        hp.reset()
        hp.map_round(order[0], len(order) == 1)
        if len(order) > 1:
            hp.map_rounds(order[1:], True)
        st2, cat2, act2 = hp.download()
        assert st0.tobytes() == st2.tobytes() and (act0 == act2).all() and (cat0 == cat2).all()
        hp.close()


def test_reads_of_21_seeds_take_the_wide_build(ds_long, ds_tiny2r):
    """maxReadLength 300 at k = 14 .. 18 needs more than the 16 seeds per read of the default kernels (round 1 returned CM_ELIMIT):
    cm_create picks the 24-seed build of the kernels; all rounds equal the oracle; cm_chain_batch (16-fragment ABI) declines."""
    P = cl.default_params(kmer=14)
    assert P.max_read_len // P.kmer == 21
    _run_all_rounds(ds_long, P)
    _run_all_rounds(ds_long, cl.default_params(kmer=14, scan_level=2, max_ed=6, seed_lim=200))
    hp = cl.HotPath(P)
    hp.load_contig(0, ds_long.hi.views[0], ds_long.hi.annots[0])
    hp.upload(ds_long.batch)
    with pytest.raises(RuntimeError):
        hp.chains(0)
    a1, b1, c1, S = hp.seeds(0)
    assert S == 21
    a0, b0, c0 = op.seeds(P, ds_long.ohi.views[0], ds_long.batch, S)
    assert (c0 == c1).all() and (b0 == b1).all() and (a0[c0 > 0] == a1[c0 > 0]).all()
    hp.close()
    # the same context type with short reads: a k = 20 index and max_read_len 400 (20 seeds) also lands in the wide build
    _run_all_rounds(ds_tiny2r, cl.default_params(kmer=ds_tiny2r.kmer, max_read_len=400))
    # beyond 24 seeds is the documented limit
    with pytest.raises(RuntimeError):
        hp = cl.HotPath(cl.default_params(kmer=14, max_read_len=400))
        hp.upload(cl.ReadBatch(np.full((2, 380), ord("A"), np.uint8), np.full((2, 380), ord("C"), np.uint8)))


def test_extension_memo_limit_is_recovered(ds_tiny, tmp_path, monkeypatch):
    """The three cases of test_hostemu_parity.test_extension_memo_limit on the device: a full memo alone and a short end piece
    alone equal the oracle in the first pass; both in one extend call make the pair kernels leave the pair untouched and queue
    it for the re-run launch (spill area behind the 8 memo entries): all three equal the oracle, no CM_ELIMIT -- the reference's
    memo is an unbounded std::map (src/extend.cpp:299,375), one pair must not fail a batch.  The same pair inside a batch of
    ordinary pairs, in the light and (CM_HEAVY_COST=0: every pair) the heavy kernel."""
    from test_hostemu_parity import tiny_exon_case
    from conftest import _Shim
    P = cl.default_params(max_ed=6)
    _run_all_rounds(tiny_exon_case(tmp_path, 10, 0), P)
    _run_all_rounds(tiny_exon_case(tmp_path, 6, 4), P)
    both = tiny_exon_case(tmp_path, 10, 4)
    _run_all_rounds(both, P)
    # that pair (several copies) among 300 ordinary pairs drawn from the same contig
    from circminer_amd import synth
    chrom = both.d.contigs[0]
    rng = np.random.default_rng(4)
    st = rng.integers(100, len(chrom) - 600, 300)
    ar = np.arange(150)
    g1 = chrom[st[:, None] + ar]
    g2 = synth.revcomp(chrom[(st + 200)[:, None] + ar])
    s1 = np.concatenate([g1, np.repeat(both.batch.seq1.reshape(1, -1), 5, 0)])
    s2 = np.concatenate([g2, np.repeat(both.batch.seq2.reshape(1, -1), 5, 0)])
    order = rng.permutation(len(s1))
    mixed = _Shim(both, cl.ReadBatch(s1[order], s2[order]))
    st_mixed = _run_all_rounds(mixed, P)
    assert np.isin(st_mixed["type"], [cl.CAT["CONCRD"], cl.CAT["CONGNM"]]).sum() >= 295       # genomic pairs: CONGNM; the spliced copies: CONCRD
    monkeypatch.setenv("CM_HEAVY_COST", "0")
    _run_all_rounds(mixed, P)


def test_rerun_launch_under_a_two_entry_memo(tmp_path):
    """A second build of the library with a 2-entry extension memo that flags every dropped insert (-DCM_MEMO_N=2
    -DCM_MEMO_STRICT): a large share of the spliced pairs now goes through the re-run launch of k_pair (spill area in global
    memory, pairs skipped by both pair kernels, nothing written in the first pass).  The parity suites of this file that map
    whole data sets through all rounds must still equal the oracle bit for bit.  Runs them in ONE child process (the library
    is chosen at import time through CM_LIB)."""
    import subprocess
    import sys
    from circminer_amd import _build
    so = _build.build(tag="memo2", flags=["-DCM_MEMO_N=2", "-DCM_MEMO_STRICT"])
    env = dict(os.environ, CM_LIB=so, CM_EXPECT_RERUNS="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"), "-k",
                        "test_map_parity_all_rounds or test_map_parity_param_variants or test_ragged_and_dirty_reads or test_rounds_in_one_call "
                        "or test_multi_tile_batches or test_staged_batches_overlap or test_config5_stress_params "
                        "or test_extension_memo_limit_is_recovered or test_reruns_are_counted"],
                       env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    import re
    m = re.search(r"(\d+) passed", r.stdout)
    # 4 + 3 + 1 + 2 + 1 + 1 + 1 + 1 + 1: the selection must not silently shrink when tests are renamed
    assert m and int(m.group(1)) >= 15, r.stdout[-1500:]


def test_heavy_pipeline_fall_back_and_variants():
    """The heavy pairs of a tile go through a pipeline of full-width kernels whose arrays hold a fixed number of mate-pair tasks
    and unpaired chains; a pair that does not fit is mapped whole by k_pair_heavy, launched late (when the stage is settled and
    the list is known to hold something) with several tiles, at once with one.  With room for 40 tasks and 24 unpaired chains
    most pairs take that way (CM_HEAVY_COST=2: every pair with chains on both reads counts as heavy); with room for 3 unpaired chains the
    pairs that get as far as those do.  Also: the second attempt through the fall-back kernel (CM_HP_ATTEMPTS=1), and the pipeline
    switched off (the round-3 path).  Every variant must equal the oracle on the whole-data-set suites; ONE child process each
    (the knobs are read once per process)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for knobs in ({"CM_HP_TASKS_CAP": "40", "CM_HP_UNP_CAP": "24", "CM_HEAVY_COST": "2"}, {"CM_HP_TASKS_CAP": "100000", "CM_HP_UNP_CAP": "3", "CM_HEAVY_COST": "2"}, {"CM_HP_ATTEMPTS": "1"}, {"CM_HEAVY_PIPELINE": "0"}):
        env = dict(os.environ, **knobs)
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"), "-k",
                            "test_map_parity_all_rounds or test_rounds_in_one_call or test_multi_tile_batches or test_ragged_and_dirty_reads"],
                           env=env, capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, str(knobs) + r.stdout[-3000:] + r.stderr[-2000:]
        m = re.search(r"(\d+) passed", r.stdout)
        assert m and int(m.group(1)) >= 7, str(knobs) + r.stdout[-1500:]


def test_reruns_are_counted(ds_small):
    """cm_prof_counters [4] counts the pair-rounds the re-run launch mapped: zero for the product
    build on ordinary data, many under the 2-entry strict test build (CM_EXPECT_RERUNS, see the test above)."""
    P = cl.default_params()
    hp = cl.HotPath(P)
    hp.prof(True)
    hp.load_contig(0, ds_small.hi.views[0], ds_small.hi.annots[0])
    hp.upload(ds_small.batch)
    hp.map_round(0, True)
    hp.sync()
    reruns = hp.prof_get()[2][4]
    hp.close()
    if os.environ.get("CM_EXPECT_RERUNS"):
        assert reruns > 100, reruns
    else:
        assert reruns == 0


@pytest.mark.parametrize("tile", [None, "300"])
def test_cross_batch_prefetch_is_used_and_discarded_correctly(ds_tiny2r, ds_dirty, monkeypatch, tile):
    """cm_map_rounds seeds / chains the staged batch's first item (first tile, first round) under the resident batch's last pair
    stage.  Every batch equals the oracle whether that work is taken over (same first slot, same contig: launches[7] counts it)
    or has to be discarded (slot reloaded in between, another first slot, a call that is not the batch's last, a staged batch
    that does not fit the resident workspace).  tile = 300: two tiles per 600-pair batch, walked round by round (a tile's
    seeding then uses the flags its previous pair stage wrote)."""
    if tile:
        monkeypatch.setenv("CM_TILE_PAIRS", tile)
    ds = ds_tiny2r
    P = cl.default_params(kmer=ds.kmer)
    hp = cl.HotPath(P)
    for ci in range(ds.hi.n_contigs):
        hp.load_contig(ci, ds.hi.views[ci], ds.hi.annots[ci])
    n = ds.batch.n
    h = n // 2
    pa = hp.pinned_batch(ds.d.seq1[:h], ds.d.seq2[:h])
    pb = hp.pinned_batch(ds.d.seq1[h:2 * h], ds.d.seq2[h:2 * h])
    small = hp.pinned_batch(ds.d.seq1[:h // 2], ds.d.seq2[:h // 2])
    want, _, _ = op.map_all_rounds(P, ds.ohi, ds.batch)
    slots = list(range(ds.hi.n_contigs))

    def taken():
        return hp.prof_get()[1][7]

    hp.prof(True)
    hp.stage(pa); hp.swap()
    hp.stage(pb)
    hp.map_rounds(slots)                                   # prefetches B's first round
    assert hp.download()[0].tobytes() == want[:h].tobytes() and taken() == 0
    hp.swap(); hp.stage(pa)
    hp.map_rounds(slots)                                   # takes it over; prefetches A's
    assert hp.download()[0].tobytes() == want[h:2 * h].tobytes() and taken() == 1
    hp.swap(); hp.stage(pb)
    hp.load_contig(0, ds.hi.views[0], ds.hi.annots[0])    # slot 0 reloaded: the prefetched chains are stale
    hp.map_rounds(slots)
    assert hp.download()[0].tobytes() == want[:h].tobytes() and taken() == 1
    hp.swap(); hp.stage(pa)                                # B resident (prefetched under the previous call)
    hp.map_rounds(slots[::-1], last_is_final=False)        # another first slot: discarded; not the last call: no prefetch
    hp.reset()
    hp.map_rounds(slots)
    assert hp.download()[0].tobytes() == want[h:2 * h].tobytes() and taken() == 1
    hp.swap(); hp.stage(small)
    hp.map_rounds(slots)                                   # A again, taken over; prefetches the smaller batch
    assert hp.download()[0].tobytes() == want[:h].tobytes() and taken() == 2
    hp.swap(); hp.stage(pb)                                # pb is larger than the resident batch now: no prefetch
    hp.map_rounds(slots)
    assert hp.download()[0].tobytes() == want[:h // 2].tobytes() and taken() == 3
    hp.swap(); hp.stage(ds_dirty.batch)
    hp.map_rounds(slots)
    assert hp.download()[0].tobytes() == want[h:2 * h].tobytes() and taken() == 3
    hp.close()
