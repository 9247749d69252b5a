"""A/B/... of several builds of the library on the bench's step, in ONE process on ONE box: the workload (genome, index, reads) is
built once, then every library maps the same batches, round-robin, REPS times.  Boxes of the pool differ by 10 - 40 % in kernel
times from call to call, so numbers of different gpurun calls do not compare; numbers of one call of this script do.
usage: ab_multi.py name=path.so [name=path.so ...]      env: PAIRS (2^21), STEPS (10), REPS (2), WORKLOAD (hg38like)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from circminer_amd import lib as cl, synth
libs = [a.split("=", 1) for a in sys.argv[1:]]          # name=path.so[@ENV=VALUE[,ENV=VALUE]]: the variables are set while that library runs
pairs = int(os.environ.get("PAIRS", 1 << 21)); steps = int(os.environ.get("STEPS", "10")); reps = int(os.environ.get("REPS", "2"))
wl = os.environ.get("WORKLOAD", "hg38like")
d = synth.generate(wl, n_pairs=2 * pairs, seed=38)
open("/tmp/ab.gtf", "w").write(d.gtf_text)
hi = cl.HostIndex(d.contigs, d.chr_table, "/tmp/ab.gtf", kmer=20, n_threads=max(1, (os.cpu_count() or 8)))
P = cl.default_params()
default_load = cl.load
KN = ["k_seed", "k_chain", "k_pair", "k_scan", "k_pair_heavy", "k_classify", "k_chain_heavy"]
res = {}
for rep in range(reps):
    for name, path in libs:
        path, _, envs = path.partition("@")
        envs = dict(e.split("=", 1) for e in envs.split(",") if e)
        os.environ.update(envs)              # (a library reads its knobs once, when first used: give every setting its own copy of the .so)
        cl._lib = None
        L = default_load(os.path.abspath(path))
        cl.load = lambda path=None, L=L: L
        hp = cl.HotPath(P)
        for ci in range(hi.n_contigs):
            hp.load_contig(ci, hi.views[ci], hi.annots[ci])
        batches = [hp.pinned_batch(d.seq1[i * pairs:(i + 1) * pairs], d.seq2[i * pairs:(i + 1) * pairs]) for i in range(2)]
        turn = 0
        def step():
            global turn
            turn ^= 1
            hp.stage(batches[turn]); hp.map_rounds(list(range(hi.n_contigs)), True); r = hp.collect_records(0); hp.swap(); return r
        hp.stage(batches[0]); hp.swap(); hp.prof(True)
        for _ in range(6): step()
        hp.sync(); hp.prof_reset()
        t = time.perf_counter()
        for _ in range(steps): rec = step()
        hp.sync(); dt = time.perf_counter() - t
        ms, launches, counters = hp.prof_get()
        st = hp.download()[0]
        import hashlib
        dig = hashlib.sha1(st.tobytes()).hexdigest()[:12]
        print("%-10s rep %d: %6.2f M pairs/s  %6.2f ms/step   per launch: %s   records %d  state %s" % (
            name, rep, pairs * steps / dt / 1e6, dt / steps * 1e3, " ".join("%s %.2f" % (KN[i][2:], ms[i] / max(launches[i], 1)) for i in (0, 1, 6, 2, 4, 5)), len(rec), dig), flush=True)
        res.setdefault(name, []).append(dt / steps * 1e3)
        hp.close(); del hp, batches
        for k in envs: os.environ.pop(k, None)
for name, v in res.items():
    print("%-10s best %.2f ms/step  mean %.2f" % (name, min(v), sum(v) / len(v)))
