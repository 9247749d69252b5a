"""N>1 path on CPU: world_size-2 gloo run of the sharding + BSJ gatherv used by bench.py.
Each rank maps its contiguous shard (CPU oracle stands in for the per-rank mapper here: this test is
about the sharding and the collective, the kernels are covered by the gpu tests)."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from circminer_amd import dist as cdist, lib as cl, synth
    from oracle import oracle_py as op
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    d = synth.generate("tiny", n_pairs=n_total, seed=21)
    gtf = os.path.join(outdir, f"r{rank}.gtf")
    open(gtf, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf)
    a, b = cdist.shard_bounds(n_total, rank, world)
    batch = cl.ReadBatch(d.seq1[a:b], d.seq2[a:b])
    P = cl.default_params()
    st, act, _ = op.map_all_rounds(P, hi, batch)
    idx = np.nonzero(act)[0].astype(np.uint64)
    mine = cdist.pack_records(idx + a, st[idx])
    rec = cdist.gather_bsj(mine)
    assert (rec is None) == (rank != 0)
    # the stepping form bench.py uses: same buffers reused over several batches, an empty batch in between
    g = cdist.BsjGather(len(mine))
    g.submit(g.fill(mine[:len(mine) // 2]))
    half = g.result()
    half = None if half is None else half.copy()
    g.submit(g.fill(mine[:0]) if rank == 0 else g.fill(mine[:1]))
    one = g.result()
    one = None if one is None else one.copy()
    g.submit(g.fill(mine))
    again = g.result()
    if rank == 0:
        assert again.tobytes() == rec.tobytes() and len(one) == 1 and one["pair"][0] >= b
        assert (np.diff(rec["pair"].astype(np.int64)) > 0).all()
        np.save(os.path.join(outdir, "gathered.npy"), rec)
        np.save(os.path.join(outdir, "half.npy"), half)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover():
    from circminer_amd import dist as cdist
    for n, w in ((10, 3), (1000, 8), (7, 8), (0, 2)):
        cuts = [cdist.shard_bounds(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))


def test_two_rank_gather_equals_single_process(ds_tiny):
    import torch.multiprocessing as mp
    from circminer_amd import dist as cdist, lib as cl
    from oracle import oracle_py as op
    n_total = 600
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_worker, args=(2, _free_port(), n_total, td), nprocs=2, join=True)
        got = np.load(os.path.join(td, "gathered.npy"))
    # single-process answer on the same reads (generator is seed-deterministic)
    from circminer_amd import synth
    d = synth.generate("tiny", n_pairs=n_total, seed=21)
    batch = cl.ReadBatch(d.seq1, d.seq2)
    st, act, _ = op.map_all_rounds(cl.default_params(), ds_tiny.hi, batch)
    idx = np.nonzero(act)[0].astype(np.uint64)
    want = cdist.pack_records(idx, st[idx])
    assert len(got) == len(want) and got.tobytes() == want.tobytes()


def _file_worker(rank, world, port, td):
    """One rank of a sharded stage 1 on files: its block of the FASTQ (cm_fastq_open_shard) -> mapped (CPU oracle standing in for
    the device) -> .part<rank> remain files (cm_writer) -> barrier -> rank 0 merges the parts and runs stage 2 (cm_circ_run)."""
    sys.path.insert(0, ROOT)
    import pickle
    import torch.distributed as dist
    from circminer_amd import lib as cl
    from oracle import oracle_py as op
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    job = pickle.load(open(os.path.join(td, "job.pkl"), "rb"))
    hi = op.OracleIndex(job["contigs"], job["chr_table"], job["gtf"])
    P = cl.default_params()
    rd = cl.FastqReader(job["fq1"], job["fq2"], job["chr_table"], P.max_ed, rank=rank, world=world)
    rounds = hi.n_contigs
    part = f".part{rank}" if world > 1 else ""                    # cm_mapping_run's naming
    w = cl.RecordWriter(f"{job['out']}_{rounds}_remain_R1.fastq{part}", f"{job['out']}_{rounds}_remain_R2.fastq{part}", job["chr_table"])
    while True:
        b = rd.next_batch(257)                                     # several batches per rank
        if b is None:
            break
        o1 = np.ctypeslib.as_array(b.c.off1, (b.n + 1,)).copy()
        o2 = np.ctypeslib.as_array(b.c.off2, (b.n + 1,)).copy()
        rb = cl.ReadBatch(np.ctypeslib.as_array(b.c.seq1, (int(o1[-1]),)).copy(), np.ctypeslib.as_array(b.c.seq2, (int(o2[-1]),)).copy(),
                          np.diff(o1).astype(np.int64), np.diff(o2).astype(np.int64))
        st, act, _ = op.map_all_rounds(P, hi, rb)
        w.write_remain(b, st, np.nonzero(act)[0])
    w.close()
    rd.close()
    dist.barrier()
    if rank == 0:
        cl.merge_parts(job["out"], rounds, world, report=0)
        cl.run_circ(job["idx"], job["gtf"], job["out"], rounds, cl.default_params(kmer=0))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_end_in_one_circ_report(built, tmp_path):
    """SURVEY 8(e) end to end on files: two rank processes each take their block of the paired FASTQ, write their part of the
    last round's remain files, rank 0 concatenates the parts and runs stage 2 once.  The remain files, candidates.pam and
    circ_report are the bytes of a one-process run (the blocks are contiguous and rows are written in input order; stage 2 sorts
    by gspos anyway, src/process_circ.cpp:188)."""
    import pickle
    import torch.multiprocessing as mp
    from circminer_amd import lib as cl, synth
    from oracle import oracle_py as op
    from stage2_util import write_fastq_pair
    d = synth.generate("tiny2r", n_pairs=2400, seed=9, mix=(0.4, 0.2, 0.4))
    td = str(tmp_path)
    gtf = os.path.join(td, "a.gtf")
    open(gtf, "w").write(d.gtf_text)
    fa = os.path.join(td, "ref.fa")
    with open(fa, "w") as f:
        for name, con, start, ln in d.chr_table:
            f.write(f">{name}\n{d.contigs[con - 1][start:start + ln].tobytes().decode()}\n")
    packed, info = cl.pack_genome(fa, 150_000)
    idx = cl.write_index(packed, kmer=20, n_threads=4)
    fq1, fq2 = write_fastq_pair(td, d, 2400)
    outs = {}
    for world in (1, 2):
        out = os.path.join(td, f"w{world}")
        pickle.dump(dict(contigs=d.contigs, chr_table=d.chr_table, gtf=gtf, fq1=fq1, fq2=fq2, out=out, idx=idx), open(os.path.join(td, "job.pkl"), "wb"))
        mp.spawn(_file_worker, args=(world, _free_port(), td), nprocs=world, join=True)
        outs[world] = {s: open(out + s, "rb").read() for s in ("_2_remain_R1.fastq", "_2_remain_R2.fastq", ".candidates.pam", ".circ_report")}
        assert not [f for f in os.listdir(td) if ".part" in f]
    assert outs[1] == outs[2]
    assert outs[2][".circ_report"].count(b"\n") > 20 and outs[2]["_2_remain_R1.fastq"].count(b"\n") > 400


def test_bench_gpus_n_starts_n_ranks_and_fails_cleanly_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent starts two fresh rank processes (RANK / WORLD_SIZE / MASTER_* set,
    nothing GPU-related touched in the parent); on this GPU-less box both ranks join the process group (gloo rehearsal), find no
    HIP device, leave the group and exit non-zero -- no hang, no CPU fallback, no JSON line."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the CPU-only rehearsal")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs a HIP device") >= 2 and "rank exit codes [1, 1]" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # a launcher-provided WORLD_SIZE that disagrees with --gpus is refused up front
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr
