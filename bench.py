#!/usr/bin/env python3
"""Benchmark of the MI355X mapping hot path (BASELINE.json metric: paired reads/s).

  python bench.py --gpus N --steps K --warmup W            (N > 1: starts N rank processes itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload = the configuration the metric is quoted on (BASELINE.json configs[2], scaled to one step): an hg38-sized
synthetic genome (24 chromosomes with hg38's lengths, 3.09 Gbp -> three packed contigs, all resident in HBM, three mapping
rounds), k = 20, batches of 2^22 2x150 bp pairs = two launch tiles of 2^21 (with two tiles the library walks them round by
round, so a tile's seeding sees the flags its previous pair stage wrote), eight distinct batches taking turns.

A "step" is one batch through the whole hot path:
  * its reads come from (page-locked) host memory: cm_reads_stage copies batch k+1 over PCIe on a copy stream while batch
    k is mapped -- the H2D copy is inside the timed region, one per step;
  * every mapping round (one per packed contig) of the resident batch;
  * the back-splice-junction hand-off (cm_collect_records; N > 1: device records gathered to rank 0 over RCCL).
The index, genome and annotation are staged into HBM once, before the timed region (they are the reference's loaded
hash table).

Weak scaling: every rank maps its own stream of `--pairs`-pair batches against its own replica of the index;
value = (pairs of all ranks x steps) / max-over-ranks time.  The host-side index is built once (rank 0) and shared with
the other ranks through memory-mapped files.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# one context drives eight HIP streams at once (cm_map_rounds: the pair stage of a round overlaps the seeding / chaining of the next,
# plus the H2D staging copy); the runtime's default of 4 hardware queues would make two of them share a queue.  Must be set
# before anything initialises HIP (torch does).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
def tile_pairs(n):
    """pairs per launch tile of a batch of n pairs (cm_hot.hip tile_for)"""
    if os.environ.get("CM_TILE_PAIRS"):
        return min(n, int(os.environ["CM_TILE_PAIRS"]))
    if n <= 2 << 20:
        return min(n, 1 << 20)
    return min(((n + 1) // 2 + 65535) // 65536 * 65536, 1 << 21)


KERNELS = ["k_seed", "k_chain", "k_pair", "k_scan", "k_pair_heavy", "k_classify", "k_chain_heavy"]
METRIC = "paired reads/sec (whole node), hg38 k=20, 2x150 bp; circ_report bit-exact"
WORKLOAD_NOTE = {
    "hg38like": "BASELINE.json configs[2], SURVEY 8(d) preset: hg38-sized synthetic genome, tiered repeat families",
    "hg38like_sparse": "BASELINE.json configs[2] layout, rounds 1-2 genome: hg38-sized, 6 x 100 000-copy repeat families, sparse genes",
    "chr21": "BASELINE.json configs[1]: chr21-like synthetic contig",
}


def algorithmic_bytes(counters):
    """SURVEY.md 8(d) per-pair-round figure, split by the kernel that moves the bytes
    (DESIGN.md 5): probes pay 16 B (bucket offset + count header) + 8 B per binary-search touch
    and the two reads come in once (300 B); chaining consumes 8 B per retained hit; pairing /
    extension is charged the survey's upper bound of four 170-byte reference windows (1360 B)
    plus the 96-byte result record."""
    probes, touches, hits, pair_rounds = counters[:4]
    return [16 * probes + 8 * touches + 300 * pair_rounds, 8 * hits, (1360 + 96) * pair_rounds, 0, 0, 0, 0]


def cpu_baseline(P, hi, batch, target_s=12.0):
    """Oracle (CPU restatement, kind 'port') on a bounded sample of the same workload, all host
    cores of this box (ctypes releases the GIL).  Reported, not the target."""
    from oracle import oracle_py as op
    cores = max(1, os.cpu_count() or 1)            # every logical CPU of the box ("all host cores")
    probe_n = min(batch.n, 4000)
    st, act = op.default_state(P, batch.n)
    t = time.time()
    for ci in range(hi.n_contigs):
        op.map_round(P, hi.views[ci], hi.annots[ci], batch, ci == hi.n_contigs - 1, st, act, 0, probe_n)
    rate1 = probe_n / max(time.time() - t, 1e-6)
    n = int(min(batch.n, max(probe_n, rate1 * cores * target_s * 0.6)))
    bounds = [(n * i) // cores for i in range(cores + 1)]
    st0, act0 = op.default_state(P, batch.n)
    fresh_st, fresh_act = st0.copy(), act0.copy()

    def work(a, b, reps):
        for _ in range(reps):
            st0[a:b] = fresh_st[a:b]               # every pass starts from the first-round state
            act0[a:b] = fresh_act[a:b]
            for ci in range(hi.n_contigs):
                op.map_round(P, hi.views[ci], hi.annots[ci], batch, ci == hi.n_contigs - 1, st0, act0, a, b)

    def run(reps):
        th = [threading.Thread(target=work, args=(bounds[i], bounds[i + 1], reps)) for i in range(cores)]
        t = time.time()
        for x in th:
            x.start()
        for x in th:
            x.join()
        return time.time() - t

    dt1 = run(1)                                    # one pass sizes the sample: about target_s seconds of wall time in all
    reps = max(1, min(40, int(round((target_s - dt1) / max(dt1, 1e-3)))))
    dt = run(reps)
    cpu_baseline.sample = (n, st0[:n].copy(), act0[:n].copy())      # the oracle's final states: what the line's parity check compares
    return {"value": n * reps / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"first {n} pairs of one batch of the same workload x {reps} passes, all {hi.n_contigs} rounds, "
                      f"oracle/cm_oracle.cpp on {cores} threads, {dt:.1f}s",
            "single_thread_value": rate1,
            "note": "kind 'port': the reference itself cannot be built in this image (its lib/ submodules are absent), so there is "
                    "no reference --thread N run and no reference/port bridge ratio"}


def write_fastq_fixed(path, seqs, mate, first=0, chunk=1 << 19, append=False):
    """(n, L) uint8 reads -> FASTQ text with fixed-width names "@r<9 digits>/<mate>" and constant qualities, written in
    vectorised chunks (a Python loop over 8 M records would take minutes)."""
    n, L = seqs.shape
    rec = 14 + L + 3 + L + 1
    with open(path, "ab" if append else "wb") as f:
        for a in range(0, n, chunk):
            b = min(n, a + chunk)
            m = np.empty((b - a, rec), dtype=np.uint8)
            m[:, 0] = ord("@")
            m[:, 1] = ord("r")
            idx = np.arange(first + a, first + b, dtype=np.int64)
            for k in range(9):
                m[:, 2 + k] = (idx // 10 ** (8 - k)) % 10 + 48
            m[:, 11] = ord("/")
            m[:, 12] = 48 + mate
            m[:, 13] = 10
            m[:, 14:14 + L] = seqs[a:b]
            m[:, 14 + L] = 10
            m[:, 15 + L] = ord("+")
            m[:, 16 + L] = 10
            m[:, 17 + L:17 + 2 * L] = ord("I")
            m[:, 17 + 2 * L] = 10
            m.tofile(f)


def raw_read_seconds(path, n_threads):
    """Wall time to pull every byte of a file through pread() on n_threads threads (the floor of any loader of it)."""
    size = os.path.getsize(path)
    fd = os.open(path, os.O_RDONLY)
    blk = 64 << 20

    def work(t):
        buf = bytearray(blk)
        a, b = size * t // n_threads, size * (t + 1) // n_threads
        while a < b:
            got = os.preadv(fd, [memoryview(buf)[:min(blk, b - a)]], a)
            if got <= 0:
                break
            a += got

    th = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    t0 = time.time()
    for x in th:
        x.start()
    for x in th:
        x.join()
    os.close(fd)
    return time.time() - t0


def end_to_end(d, workload, n_pairs, n_threads, batch_pairs, dev_index, workdir=None):
    """What a user of the reference runs, on files (src/circminer.cpp:98-352): <ref>.packed.fa(.index, .index.info) + GTF + two
    plain-text FASTQ files -> cm_mapping_run (index file + GTF -> HBM, FASTQ text -> last round's remain files: the reference's
    default output, reportMapping = DISCARDMAPREPORT, src/commandline_parser.cpp:26) -> cm_circ_run (sort, stage 2) ->
    <out>.candidates.pam + <out>.circ_report.  Everything lives in tmpfs (no disk in the measurement).  Outside the timed
    hot-path region of the bench line; reported next to it."""
    from circminer_amd import lib as cl
    base = workdir or share_dir(f"cm_e2e_{os.getpid()}", 12 * sum(len(c) for c in d.contigs) + 700 * n_pairs)
    shutil.rmtree(base, ignore_errors=True)
    os.makedirs(base)
    res = {"workload": workload, "pairs": int(n_pairs), "threads": n_threads, "batch_pairs": int(batch_pairs), "dir": os.path.dirname(base)}
    try:
        t = time.time()
        packed = os.path.join(base, "ref.fa.packed.fa")
        with open(packed, "wb") as f:                       # GenomePacker::pack_genome's output, straight from the generator's contigs
            for ci, c in enumerate(d.contigs):
                f.write(b">%d\n" % (ci + 1))
                np.ascontiguousarray(c).tofile(f)
                f.write(b"\n")
        with open(packed + ".index.info", "w") as f:
            for name, con, start, ln in d.chr_table:
                f.write(f"{con}\t{start}\t{start + ln}\t{name}\n")
        gtf = os.path.join(base, "ref.gtf")
        with open(gtf, "w") as f:
            f.write(d.gtf_text)
        fq = [os.path.join(base, f"reads_{m}.fq") for m in (1, 2)]
        have = d.seq1.shape[0]                              # more pairs than were generated: the same reads again under new names
        for m, arr in ((0, d.seq1), (1, d.seq2)):
            with open(fq[m], "wb"):
                pass
            for a in range(0, n_pairs, have):
                write_fastq_fixed(fq[m], arr[:min(have, n_pairs - a)], m + 1, first=a, append=True)
        res["prep_files_s"] = round(time.time() - t, 1)
        res["fastq_bytes"] = os.path.getsize(fq[0]) + os.path.getsize(fq[1])
        t = time.time()
        idx = cl.write_index(packed, kmer=20, n_threads=n_threads)          # `circminer --index`: once per genome, not part of a run
        res["write_index_s"] = round(time.time() - t, 1)
        res["index_bytes"] = os.path.getsize(idx)
        res["index_raw_read_s"] = round(raw_read_seconds(idx, min(n_threads, 32)), 2)      # pread()s side by side, nothing decoded
        out = os.path.join(base, "run")
        t = time.time()
        st = cl.run_mapping(idx, gtf, fq[0], fq[1], out, cl.default_params(kmer=0, device=dev_index), report=0, n_threads=n_threads,
                            batch_pairs=batch_pairs)
        stage1_s = time.time() - t
        t = time.time()
        cs = cl.run_circ(idx, gtf, out, st.rounds, cl.default_params(kmer=0), n_threads=n_threads)
        stage2_s = time.time() - t
        res.update({
            "load_s": round(st.seconds_load, 2),                      # index file + GTF -> HBM (all packed contigs resident)
            "index_GBps": round(res["index_bytes"] / max(st.seconds_load, 1e-9) / 1e9, 2),
            "map_s": round(st.seconds_map, 3),                        # FASTQ text -> remain files
            "fastq_to_remain_pairs_per_s": st.pairs / max(st.seconds_map, 1e-9),
            "map_parts_s": {"parse": round(st.seconds_parse, 3), "device": round(st.seconds_device, 3), "write": round(st.seconds_write, 3)},
            "bsj_pairs": int(st.bsj_pairs), "stage2_s": round(stage2_s, 2), "candidate_rows": int(cs.candidate_rows), "calls": int(cs.calls),
            "stage1_total_s": round(stage1_s, 2),
            "fastq_to_circ_report_pairs_per_s_excl_load": st.pairs / max(st.seconds_map + stage2_s, 1e-9),
            "fastq_to_circ_report_pairs_per_s_incl_load": st.pairs / max(stage1_s + stage2_s, 1e-9),
            "circ_report_rows": sum(1 for _ in open(out + ".circ_report")),
        })
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent never touches the GPU),
    pass rank 0's JSON line through."""
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"bench.py: rank exit codes {codes}")
    line = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    if not line or json.loads(line[-1]).get("n_gpus") != n:
        raise SystemExit(f"bench.py: rank 0 did not report n_gpus == {n}")


def share_dir(tag, need_bytes):
    """Directory for the index arrays shared between the ranks of one node: /dev/shm when it has the room, else TMPDIR."""
    for base in ("/dev/shm", tempfile.gettempdir()):
        try:
            if shutil.disk_usage(base).free > need_bytes * 1.1:
                return os.path.join(base, tag)
        except OSError:
            pass
    return os.path.join(tempfile.gettempdir(), tag)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="hg38like", choices=sorted(WORKLOAD_NOTE))
    ap.add_argument("--pairs", type=int, default=1 << 22, help="pairs per batch (= per step and GPU); the library maps a batch in tiles of <= 2^21 pairs, "
                    "two for the default (cm_hot.hip tile_for)")
    ap.add_argument("--seed", type=int, default=38)
    ap.add_argument("--batches", type=int, default=8, help="distinct batches of --pairs pairs that take turns in the timed region (>= 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e", type=int, default=0, metavar="PAIRS",
                    help="also run the file-to-file flow (index file + GTF + FASTQ text -> circ_report) on this many pairs and add an "
                         "'end_to_end' object to the line (minutes of extra preparation at hg38 scale: off by default)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (gloo: CPU rehearsal of the launch, tests only)")
    ap.add_argument("--traffic", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="per-launch HBM bytes from a separate rocprofv3 --pmc pass, if collected")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])

    # stdout carries the one JSON line and nothing else: whatever native libraries print there while we run (RCCL's version
    # banner at communicator creation, for one) is sent to stderr; fd 1 is restored just before the line is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 harness on a box with ONE GPU (rank start-up, index sharing through memory-mapped files, barriers,
    # max-over-ranks timing, record gather): `--backend gloo` + CM_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and gathers the
    # records through host tensors.  Not a measurement: the ranks share the card.
    one_device = bool(os.environ.get("CM_BENCH_ONE_DEVICE"))
    dev_index = 0 if one_device else local_rank
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    dist = None
    # CM_BENCH_FORCE_DIST=1: take the N>1 code path (process group, gatherv, barriers) with a single rank -- a rehearsal of
    # the RCCL calls on a one-GPU box
    multi = world > 1 or bool(os.environ.get("CM_BENCH_FORCE_DIST"))
    if multi:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(minutes=30))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
    if not torch.cuda.is_available():
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    dev = torch.device("cuda", dev_index)
    gather_dev = dev if args.backend == "nccl" else torch.device("cpu")

    import __graft_entry__ as ge
    if multi:                           # one rank compiles (if anything is stale), the others wait: no concurrent writes of the .o / .so
        if rank == 0:
            ge.build()
        dist.barrier()
    ge.build()
    from circminer_amd import dist as cdist, lib as cl, synth

    # ---- synthetic workload (same genome + annotation on every rank, own stream of reads) ----
    t0 = time.time()
    n_threads = max(1, (os.cpu_count() or 8) // max(world, 1))
    # two different batches take turns, so that consecutive steps really move different reads over PCIe
    d = synth.generate(args.workload, n_pairs=max(2 * args.pairs, min(args.e2e, 1 << 23) if rank == 0 else 0), seed=args.seed, read_seed=rank)
    gen_s = time.time() - t0
    tag = f"cm_bench_{args.workload}_{args.seed}_{os.environ.get('MASTER_PORT', '0')}"
    sdir = share_dir(tag, 8 * sum(len(c) for c in d.contigs)) if world > 1 else None
    with tempfile.TemporaryDirectory() as td:
        gtf = os.path.join(td, "ref.gtf")
        with open(gtf, "w") as f:
            f.write(d.gtf_text)
        if world == 1:
            hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=n_threads)
        else:
            # the k-mer index is built once per node: rank 0 -> memory-mapped files -> the other ranks
            if rank == 0:
                shutil.rmtree(sdir, ignore_errors=True)
                hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=os.cpu_count() or 8)
                hi.save_index(sdir)
            dist.barrier()
            if rank != 0:
                hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, index_dir=sdir)
    P = cl.default_params(device=dev_index)
    prep_s = time.time() - t0
    # repeat content of the workload as a probe sees it (SURVEY 8(d) targets: ~10 % of the 20-mers with > 1 hit, ~1 % beyond seedLim)
    hit_stats = hi.hit_stats(P.seed_lim, n_threads) if rank == 0 else []

    hp = cl.HotPath(P)
    for ci in range(hi.n_contigs):
        hp.load_contig(ci, hi.views[ci], hi.annots[ci])
    if world > 1:
        dist.barrier()
        if rank == 0:
            shutil.rmtree(sdir, ignore_errors=True)          # mappings of the other ranks stay valid until they drop them
    batches = [hp.pinned_batch(d.seq1[i * args.pairs:(i + 1) * args.pairs], d.seq2[i * args.pairs:(i + 1) * args.pairs]) for i in range(2)]
    load_s = time.time() - t0 - prep_s
    # more distinct batches (the timed region streams them round-robin): own streams of reads off the same genome, made on host
    # threads; batches 0 and 1 stay what earlier rounds measured
    t_more = time.time()
    n_more = max(0, args.batches - 2)
    if n_more:
        from concurrent.futures import ThreadPoolExecutor

        def more(k):
            rng = np.random.default_rng([args.seed, 7919 + 64 * rank + k])
            s1, s2 = synth.make_reads(rng, d.chr_seqs, d.genes, args.pairs)[:2]
            return s1, s2
        with ThreadPoolExecutor(max_workers=min(n_more, max(1, n_threads // 2))) as ex:
            for s1, s2 in ex.map(more, range(n_more)):
                batches.append(hp.pinned_batch(s1, s2))
    more_s = time.time() - t_more
    base = rank * args.pairs
    gather = cdist.BsjGather(args.pairs, gather_dev) if multi else None
    turn = [0]
    trace = bool(os.environ.get("CM_BENCH_TRACE"))

    def step():
        # H2D of the next batch on the copy stream, concurrent with the rounds of the resident one
        turn[0] = (turn[0] + 1) % len(batches)
        t_stage = time.perf_counter()
        hp.stage(batches[turn[0]])
        tr = [time.perf_counter()] if trace else None
        if trace:
            print("stage (ms) %.2f   since previous step's return %.2f" % ((tr[0] - t_stage) * 1e3, (t_stage - step.t_ret) * 1e3 if hasattr(step, "t_ret") else 0.0), file=sys.stderr)
        hp.map_rounds(list(range(hi.n_contigs)), True)
        if trace:
            tr.append(time.perf_counter())
        # BSJ hand-off to stage 2: records assembled on the device.  One GPU: one small D2H.  N GPUs: gatherv to rank 0 over
        # RCCL from HBM; rank 0's D2H of the gathered records overlaps the next step's rounds and is waited for in fence().
        if multi and os.environ.get("CM_BENCH_DIST_NOGATHER"):       # diagnostic: process group up, hand-off as with one GPU
            rec = hp.collect_records(base)
        elif multi and gather_dev.type == "cpu":                     # gloo rehearsal: records through host tensors
            rec0 = hp.collect_records(base)
            gather.submit(gather.fill(rec0))
            rec = None
        elif multi:
            ptr = gather.send_ptr()
            if trace:
                tr.append(time.perf_counter())
            nrec = hp.collect_records_device(base, args.pairs, ptr)
            if trace:
                tr.append(time.perf_counter())
            gather.submit(nrec)
            rec = None
        else:
            rec = hp.collect_records(base)
        if trace:
            tr.append(time.perf_counter())
        hp.swap()
        if trace:
            tr.append(time.perf_counter())
            print("step parts (ms): map_rounds(host) %.2f" % ((tr[1] - tr[0]) * 1e3), [round((b - a) * 1e3, 2) for a, b in zip(tr[1:], tr[2:])], file=sys.stderr)
            step.t_ret = time.perf_counter()
        return rec

    def fence():
        hp.sync()
        if multi:
            gather.result()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    hp.stage(batches[0])
    hp.swap()
    hp.prof(True)                      # timers on during warm-up too: their HIP events are created once and recycled
    # Setup, not warm-up: the first launches allocate the kernels' private segments and lazily created runtime
    # objects, and the device shows one ~7 ms stall within its first ~100 ms of work (tests/diag/step_times.py);
    # prime until that is behind us so that the W warm-up and K timed steps see the steady state.
    for _ in range(4):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    hp.prof_reset()
    t = time.perf_counter()
    marks = []
    for _ in range(args.steps):
        rec = step()
        marks.append(time.perf_counter() - t)
    fence()
    dt = time.perf_counter() - t
    if multi:
        rec = gather.result()
    if os.environ.get("CM_BENCH_TRACE"):
        print("step marks (ms):", [round(m * 1e3, 2) for m in marks], "end", round(dt * 1e3, 2), file=sys.stderr)
    ms, launches, counters = hp.prof_get()
    hp.prof(False)
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total_pairs = args.pairs * world * args.steps
        value = total_pairs / dt
        ab = algorithmic_bytes(counters)
        # the pair stage = class kernels + k_pair (light pairs, one per lane) with the heavy pairs' pipeline (k_hp_*) running
        # concurrently on a second stream: class [2] is timed from the k_pair launch to the join, its algorithmic
        # bytes cover all pair-rounds; classes [4], [6] are the overlapped kernels' own times
        # (the class / counting-sort kernels [5] are not added: with the rounds pipelined their timers mostly measure waiting for
        # the dispatcher behind the next round's seeding, not work)
        stage_ms = [ms[0], ms[1], ms[2]]                # heavy kernels run on a second stream inside [1] / [2]
        dom = int(np.argmax(stage_ms))
        avg_ms = stage_ms[dom] / max(launches[dom], 1)
        achieved = (ab[dom] / max(launches[dom], 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_src = None, None
        try:
            with open(args.traffic) as f:
                tj = json.load(f)
            if tj.get("workload") == args.workload and tj.get("pairs") == tile_pairs(args.pairs) and \
                    (args.workload != "hg38like" or "dense" in tj.get("preset", "")):                  # pairs per launch (tile); not the r02 genome's
                traffic = tj.get("bytes_per_launch", {}).get(KERNELS[dom])
                traffic_src = "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (profiles/traffic.json), not this run"
        except Exception:
            traffic = None
        read_bytes = int(batches[0].seq1.size + batches[0].seq2.size + 16 * (args.pairs + 1))
        # every stage against the HBM roofline, with the counter figures BASELINE.md 4 names (HBM GB/s of the probe phase, LDS
        # behaviour of the chaining phase) and the lanes a VALU instruction of each kernel carries -- the figure that says what
        # bounds the pair stage.  Times and algorithmic bytes: this run (HIP events per stage).  Counter figures: separate
        # rocprofv3 --pmc passes (profiles/traffic.json: FETCH_SIZE + WRITE_SIZE; profiles/pmc_mix.json: SQ_* of one 2^20-pair round).
        mix = {}
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_mix.json")) as f:
                mix = json.load(f)
            if mix.get("workload") != args.workload:
                mix = {}
        except Exception:
            mix = {}
        tdet = {}
        try:
            tdet = tj.get("bytes_per_launch", {}) if traffic_src else {}
        except Exception:
            tdet = {}
        stages = {}
        for name, idx, kernels in (("seed", 0, ["k_seed"]), ("chain", 1, ["k_chain", "k_chain_heavy"]), ("pair", 2, ["k_pair", "k_hp_tasks", "k_hp_dp", "k_hp_plan", "k_pair_heavy"])):
            n_l = max(launches[idx], 1)
            t_ms = ms[idx] / n_l
            ach = (ab[idx] / n_l) / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
            tb = tdet.get(KERNELS[idx])
            stages[name] = {"kernels": kernels, "avg_launch_ms": t_ms, "algorithmic_bytes_per_launch": ab[idx] / n_l, "achieved_GBps": ach,
                            "frac_of_hbm_peak": ach / HBM_PEAK_GBS, "counter_bytes_per_launch": tb,
                            "counter_GBps": (tb / (t_ms * 1e-3) / 1e9) if (tb and t_ms > 0) else None,
                            "lanes_per_valu_inst": {k: mix.get("kernels", {}).get(k, {}).get("lanes_per_valu_inst") for k in kernels},
                            "lds_bank_conflict_rate": {k: mix.get("kernels", {}).get(k, {}).get("lds_bank_conflict_rate") for k in kernels},
                            "bound": {"seed": "hbm (random 64-byte sectors)", "chain": "hbm sectors (light) / latency + LDS (heavy)",
                                      "pair": "VALU issue under control divergence"}[name]}
        stages["counter_source"] = mix.get("source") if mix else None
        out = {
            "metric": METRIC,
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32/u8 (chain scores f64)", "data": "synthetic",
            "config": {"workload": f"{WORKLOAD_NOTE[args.workload]} ({sum(len(c) for c in d.contigs)} bp in {hi.n_contigs} packed contig(s), "
                                   f"{len(d.genes)} genes / {sum(len(g.transcripts) for g in d.genes)} transcripts; indexed 20-mers with > 1 hit "
                                   f"{'/'.join('%.1f%%' % (100.0 * m / max(n, 1)) for n, m, o, _ in hit_stats)}, beyond seedLim "
                                   f"{'/'.join('%.2f%%' % (100.0 * o / max(n, 1)) for n, m, o, _ in hit_stats)} per contig), k=20, {hi.n_contigs} mapping round(s) per batch, batches of {args.pairs} 2x150 bp pairs "
                                   f"streamed from host memory ({len(batches)} distinct batches round-robin; H2D inside the timed region, overlapped with the rounds), defaults; "
                                   f"{total_pairs} pairs in the timed region",
                       "scope": "stage 1 hot path (process_read over all rounds + BSJ hand-off); reads start in host memory, "
                                "FASTQ parsing and stage-2 circ_report are outside the timed region; parity is against the "
                                "CPU oracle (parity unpinned: the reference cannot be built here)",
                       "pairs_per_gpu_per_step": args.pairs, "total_pairs": total_pairs, "rounds": hi.n_contigs,
                       "h2d_bytes_per_step": read_bytes,
                       "bsj_records_last_step": int(len(rec)),         # all ranks' records, as gathered on rank 0
                       "genes": len(d.genes), "transcripts": sum(len(g.transcripts) for g in d.genes),
                       "multi_hit_fraction": [m / max(n, 1) for n, m, o, _ in hit_stats],
                       "beyond_seed_lim_fraction": [o / max(n, 1) for n, m, o, _ in hit_stats],
                       "world_size": world,
                       "distinct_batches": len(batches),
                       "prep_seconds": {"generate": round(gen_s, 1), "index+annotation": round(prep_s - gen_s, 1), "load_to_hbm": round(load_s, 1),
                                        "more_batches": round(more_s, 1)}},
            "roofline": {"bound": "hbm", "kernel": KERNELS[dom] + ("+k_hp_* (pair stage: light kernel launch to the join with the heavy pairs' pipeline)" if dom == 2 else ""), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": avg_ms, "launches": launches[dom],
                         "algorithmic_bytes_per_launch": ab[dom] / max(launches[dom], 1)},
            "stages": stages,
            "kernels": {KERNELS[i]: {"ms_total": ms[i], "launches": launches[i], "algorithmic_bytes": ab[i]} for i in range(7)},
            "counters": {"probes": counters[0], "search_touches": counters[1], "hits_consumed": counters[2], "pair_rounds": counters[3],
                         "pair_rounds_rerun": counters[4]},
        }
        # The line's own parity evidence (outside the timed region): batch 0 once more through the same calls the steps make --
        # staged, swapped in, all rounds in one call, two tiles -- and its final MatchedRead (72 B) + re-queue flag of every pair
        # kept; the cpu_baseline leg below maps (a prefix of) the same batch with the oracle and the two are compared byte for byte.
        hp.stage(batches[0])
        hp.swap()
        hp.map_rounds(list(range(hi.n_contigs)), True)
        gpu_st, _, gpu_act = hp.download()
        if args.e2e:
            hp.close()                                          # its HBM goes back before cm_mapping_run makes a context of its own
            # (file to file the batches are 2^21 pairs: parse / device / write of consecutive batches overlap on the host, and twice as many
            # smaller batches fill that pipeline sooner -- 12.8 against 8.5 M pairs/s with 2^22)
            out["end_to_end"] = end_to_end(d, args.workload, args.e2e, os.cpu_count() or 8, min(args.pairs, 1 << 21), dev_index)
        # rank 0's host cores, after the timed region (the other ranks are idle at the final barrier by then)
        out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(P, hi, batches[0])
        if out["cpu_baseline"] is not None:
            n_s, o_st, o_act = cpu_baseline.sample
            eq = bool(gpu_st[:n_s].tobytes() == o_st.tobytes() and (gpu_act[:n_s] == o_act).all())
            out["parity"] = {"pairs": int(n_s), "equal": eq, "rounds": hi.n_contigs,
                             "what": "final cm_mapped_read (72 bytes) + re-queue flag of the first `pairs` pairs of batch 0 after all rounds: "
                                     "HIP path (staged batch, cm_map_rounds, %d-pair tiles) vs the CPU oracle (parity unpinned: the oracle is "
                                     "pinned by planted truth and its own builders, not by reference output)" % tile_pairs(args.pairs),
                             "bsj_pairs": int(gpu_act[:n_s].sum())}
            if not eq:
                bad = np.nonzero([gpu_st[i].tobytes() != o_st[i].tobytes() for i in range(n_s)])[0]
                out["parity"]["first_differences"] = [int(x) for x in bad[:8]]
        else:
            out["parity"] = None
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    hp.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
