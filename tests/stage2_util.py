"""Shared by the CPU and GPU stage-2 tests: stage-1 states -> remain FASTQ files -> sorted -> parsed back, and the oracle's
stage-2 outputs for them (oracle/ is the checker only)."""
import os
import subprocess

import numpy as np

from circminer_amd import lib as cl
from oracle import oracle_py as op


def write_fastq_pair(tmp, d, n, prefix="in", name=lambda i: f"pair{i}"):
    paths = []
    for mate, arr in ((1, d.seq1), (2, d.seq2)):
        p = os.path.join(str(tmp), f"{prefix}_{mate}.fq")
        with open(p, "w") as f:
            for i in range(n):
                s = arr[i].tobytes().decode()
                f.write(f"@{name(i)}/{mate}\n{s}\n+\n{'I' * len(s)}\n")
        paths.append(p)
    return paths


def remain_files_from_states(tmp, d, P, states, active, rounds, out="o"):
    """What stage 1 leaves for stage 2: <out>_<R>_remain_R{1,2}.fastq of the pairs still active after the last round."""
    n = len(states)
    p1, p2 = write_fastq_pair(tmp, d, n)
    rd = cl.FastqReader(p1, p2, d.chr_table, P.max_ed)
    b = rd.next_batch(n + 10)
    prefix = os.path.join(str(tmp), out)
    r1, r2 = f"{prefix}_{rounds}_remain_R1.fastq", f"{prefix}_{rounds}_remain_R2.fastq"
    w = cl.RecordWriter(r1, r2, d.chr_table)
    w.write_remain(b, states, np.nonzero(active)[0])
    w.close()
    rd.close()
    return prefix, r1, r2


def remain_files_of_active(tmp, d, P, states, active, rounds, out="o"):
    """The same two files, written from the active pairs alone (the others never reach a remain file): for batches too large
    to push through Python text I/O as a whole.  Names carry the pair's index in the batch."""
    keep = np.nonzero(active)[0]

    class _Sub:
        seq1, seq2 = d.seq1[keep], d.seq2[keep]

    p1, p2 = write_fastq_pair(tmp, _Sub, len(keep), prefix=out + "_in", name=lambda i: f"pair{keep[i]}")
    rd = cl.FastqReader(p1, p2, d.chr_table, P.max_ed)
    b = rd.next_batch(len(keep) + 10)
    prefix = os.path.join(str(tmp), out)
    r1, r2 = f"{prefix}_{rounds}_remain_R1.fastq", f"{prefix}_{rounds}_remain_R2.fastq"
    w = cl.RecordWriter(r1, r2, d.chr_table)
    if b is not None:
        w.write_remain(b, np.ascontiguousarray(states[keep]))
    w.close()
    rd.close()
    return prefix, r1, r2


def gnu_sort(path):
    """ProcessCirc::sort_fq's pipeline itself (C locale)."""
    env = dict(os.environ, LC_ALL="C")
    subprocess.check_call(f'cat {path} | paste - - - - | sort -S 64M -k2,2n | tr "\\t" "\\n" > {path}.gnu', shell=True, env=env)
    return path + ".gnu"


def oracle_stage2(tmp, hi, d, P, sorted_r1, sorted_r2, tag="oracle"):
    rd = cl.FastqReader(sorted_r1, sorted_r2, d.chr_table, P.max_ed)
    b = rd.next_batch(1 << 30)
    cand, rep = os.path.join(str(tmp), tag + ".candidates.pam"), os.path.join(str(tmp), tag + ".circ_report")
    if b is None:
        open(cand, "w").close()
        open(rep, "w").close()
    else:
        names = [b.name(i) for i in range(b.n)]
        op.circ_run(P, hi, d.chr_table, names, b, b.prior, cand, rep)
    rd.close()
    return open(cand, "rb").read(), open(rep, "rb").read()
