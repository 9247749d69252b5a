"""Multi-GPU sharding of the mapping hot path.

`process_read` is a pure function of (pair, carried state, read-only index / annotation /
params), so reads shard across GPUs with no data-path collective: one process per GPU, each with
the full index + annotation replicated in its HBM and a contiguous block of the read pairs.

The only exchange is after the last round: the pairs whose final type is CHIBSJ / CHI2BSJ
(the records `write_read_category` hands to stage 2, reference src/circminer.cpp:395-397) are
gathered to rank 0 — a variable-length gather (all_gather of counts, then a padded all_gather of
the payload; `backend="nccl"` is RCCL over xGMI on ROCm, `gloo` on CPU for the tests).
The payload is KBs-MBs, so it is latency-bound; link bandwidth is irrelevant.
"""
from __future__ import annotations

import numpy as np

from . import lib as cl

REC_DTYPE = cl.RECORD_DTYPE        # cm_record: (global pair index, state), 80 bytes


def shard_bounds(n_total: int, rank: int, world: int):
    """Contiguous block of pairs for `rank`: [r*N/W, (r+1)*N/W)."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def pack_records(global_idx: np.ndarray, states: np.ndarray) -> np.ndarray:
    n = len(global_idx)
    raw = np.empty((n, REC_DTYPE.itemsize), dtype=np.uint8)      # two strided byte copies, not per-field assignment
    raw[:, :8] = np.ascontiguousarray(global_idx, dtype="<u8").view(np.uint8).reshape(n, 8)
    raw[:, 8:] = np.ascontiguousarray(states).view(np.uint8).reshape(n, cl.MAPPED_DTYPE.itemsize)
    return raw.reshape(-1).view(REC_DTYPE)


def gather_bsj(records: np.ndarray, device=None):
    """Variable-length gather of BSJ records to every rank (rank 0 consumes them).

    Returns the concatenation over ranks, sorted by global pair index (stage 2 sorts its input
    anyway, reference src/process_circ.cpp:179-193, so any order gives the same circ_report)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.sort(records, order="pair")
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(records)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    buf = np.zeros(mx * REC_DTYPE.itemsize, dtype=np.uint8)
    raw = records.view(np.uint8).reshape(-1)
    buf[:raw.size] = raw
    mine = torch.from_numpy(buf).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = []
    for c, p in zip(counts, parts):
        if c:
            out.append(p.cpu().numpy()[:c * REC_DTYPE.itemsize].view(REC_DTYPE))
    allrec = np.concatenate(out) if out else np.zeros(0, dtype=REC_DTYPE)
    return np.sort(allrec, order="pair")
