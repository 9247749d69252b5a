"""FASTQ ingest rate of cm_fastq_next (no GPU involved): record-by-record parser vs the chunk-parallel plain-text path."""
import os, sys, time, tempfile, shutil, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import lib as cl
n = int(os.environ.get("PAIRS", "3000000"))
batch = int(os.environ.get("BATCH", str(1 << 18)))
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
rng = np.random.default_rng(1)
seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, 150), dtype=np.int8)]
q = b"I" * 150
for mate in (1, 2):
    with open(f"{td}/r_{mate}.fq", "wb") as f:
        for i in range(n):
            f.write(b"@read%d/%d\n" % (i, mate)); f.write(seq[i].tobytes()); f.write(b"\n+\n"); f.write(q); f.write(b"\n")
print("files written", flush=True)
for env in ({"CM_FASTQ_SERIAL": "1"}, {"CM_FASTQ_THREADS": "4"}, {"CM_FASTQ_THREADS": "8"}, {"CM_FASTQ_THREADS": "16"}, {"CM_FASTQ_THREADS": "32"}):
    for k in ("CM_FASTQ_SERIAL", "CM_FASTQ_THREADS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rd = cl.FastqReader(f"{td}/r_1.fq", f"{td}/r_2.fq", [], 4)
    tot = k = 0
    t = None
    while True:
        fb = cl.FastqBatch()
        rd.L.cm_fastq_next(rd.h, batch, C.byref(fb))
        if fb.reads.n_pairs == 0:
            break
        k += 1
        if k == 4:                      # steady state: the three storage generations have been touched
            t = time.time(); tot = 0
            continue
        tot += fb.reads.n_pairs
    print(env, "steady state %.2f M pairs/s" % (tot / (time.time() - t) / 1e6), flush=True)
    rd.close()
shutil.rmtree(td)
