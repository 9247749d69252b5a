"""Multi-GPU sharding of the mapping hot path.

`process_read` is a pure function of (pair, carried state, read-only index / annotation /
params), so reads shard across GPUs with no data-path collective: one process per GPU, each with
the full index + annotation replicated in its HBM and a contiguous block of the read pairs.

The only exchange is after the last round: the pairs whose final type is CHIBSJ / CHI2BSJ
(the records `write_read_category` hands to stage 2, reference src/circminer.cpp:395-397) are
gathered to rank 0 — a variable-length gather (all_gather of counts, then a padded gather of the
payload straight from HBM; `backend="nccl"` is RCCL over xGMI on ROCm, `gloo` on CPU for the tests).
The payload is KBs-MBs, so it is latency-bound; link bandwidth is irrelevant.  Only rank 0 copies
records to the host, and that copy overlaps the next batch's mapping rounds (BsjGather).
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


class BsjGather:
    """Variable-length gather of BSJ records to rank 0 (gatherv): all_gather of the per-rank counts, then one
    `gather` of the payload padded to the largest count.  Works on whatever the process group moves: device tensors
    over RCCL (`send` lives in HBM and is filled in place by cm_collect_records_device, ranks other than 0 copy
    nothing to the host) or CPU tensors over gloo (tests).

    Rank r's records are in ascending pair order (the device compaction is a stable counting sort) and the shards are
    contiguous blocks in rank order, so the rank-order concatenation rank 0 builds is already sorted by global pair
    index -- no sort here (stage 2 sorts its input anyway, reference src/process_circ.cpp:179-193).

    submit() only enqueues rank 0's device-to-host copies; result() waits for them, so the copies of one batch overlap
    the mapping rounds of the next."""

    ISZ = REC_DTYPE.itemsize

    def __init__(self, cap: int, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.on else 1
        self.rank = dist.get_rank() if self.on else 0
        self.dev = device if device is not None else torch.device("cpu")
        self.cuda = self.dev.type == "cuda"
        self.cap = max(int(cap), 1)
        self.send = torch.zeros(self.cap * self.ISZ, dtype=torch.uint8, device=self.dev)
        self._n = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self._counts = torch.zeros(self.world, dtype=torch.int64, device=self.dev)
        self._host = None
        self._total = 0
        self._sent = self._done = None

    def send_ptr(self) -> int:
        """device address of `send`, once the previous gather has finished reading it"""
        if self._sent is not None:
            self._sent.synchronize()
        return self.send.data_ptr()

    def fill(self, records: np.ndarray) -> int:
        """host records -> send buffer (CPU / test path; the GPU path writes `send` in place)"""
        n = len(records)
        if n > self.cap:
            raise ValueError(f"{n} records > capacity {self.cap}")
        raw = self.torch.from_numpy(np.ascontiguousarray(records).view(np.uint8).reshape(-1).copy())
        self.send_ptr()
        self.send[:raw.numel()].copy_(raw)
        return n

    def submit(self, n: int) -> None:
        """`send` holds n records; start the gather to rank 0"""
        torch, dist = self.torch, self.dist
        if self._done is not None:                       # previous result still landing in the host buffer
            self._done.synchronize()
        if not self.on:
            counts = [int(n)]
        else:
            self._n.fill_(int(n))
            dist.all_gather_into_tensor(self._counts, self._n)
            counts = [int(c) for c in self._counts.tolist()]
        mx = max(counts)
        total = sum(counts)
        self._total = total
        if mx == 0:
            return
        if mx > self.cap:                                # another rank holds more than this rank's buffer was sized for
            big = torch.zeros(mx * self.ISZ, dtype=torch.uint8, device=self.dev)
            big[:int(n) * self.ISZ].copy_(self.send[:int(n) * self.ISZ])
            self.send, self.cap = big, mx
        mine = self.send[:mx * self.ISZ]
        if self.on:
            recv = torch.empty((self.world, mx * self.ISZ), dtype=torch.uint8, device=self.dev) if self.rank == 0 else None
            dist.gather(mine, list(recv.unbind(0)) if self.rank == 0 else None, dst=0)
        else:
            recv = mine.view(1, -1)
        if self.cuda:
            self._sent = torch.cuda.Event()
            self._sent.record()
        if self.rank != 0:
            return
        if self._host is None or self._host.numel() < total * self.ISZ:
            self._host = torch.empty(int(total * 1.5) * self.ISZ + 64, dtype=torch.uint8, pin_memory=self.cuda)
        off = 0
        for r, c in enumerate(counts):
            nb = c * self.ISZ
            if nb:
                self._host[off:off + nb].copy_(recv[r, :nb], non_blocking=True)
                off += nb
        if self.cuda:
            recv.record_stream(torch.cuda.current_stream())
            self._done = torch.cuda.Event()
            self._done.record()

    def result(self):
        """rank 0: the gathered records of the last submit() (a view the next submit() overwrites); other ranks: None"""
        if self.rank != 0:
            return None
        if self._done is not None:
            self._done.synchronize()
        if self._total == 0:
            return np.zeros(0, dtype=REC_DTYPE)
        return self._host.numpy()[:self._total * self.ISZ].view(REC_DTYPE)


def gather_bsj(records: np.ndarray, device=None):
    """One-shot form for host-resident records: rank 0 gets the concatenation over ranks (see BsjGather), others None."""
    g = BsjGather(len(records), device)
    g.submit(g.fill(records))
    out = g.result()
    return None if out is None else out.copy()
