"""Graph-sharded multi-GPU execution: one process per GPU, whole graphs assigned to ranks, no
data-path collective except ONE all-gather of the graph-level output gf' (RCCL over xGMI via torch.distributed's
"nccl" backend; "gloo" in the CPU tests).

The reference has no multi-device code at all; what makes this legal is that every term of a graph's edge, node
and graph update depends on that graph only (src/gngraphbatch.jl builds every broadcaster per batch slice and
NNlib.batched_mul never mixes batch indices), so graphs are independent units.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def partition_graphs(edge_counts, world_size):
    """Assign whole graphs to ranks: equal graph counts (±1), balanced by edge count (sort by E_g descending, then
    deal in a snake/boustrophedon order).  Returns a list of int64 index arrays (original graph ids, ascending)."""
    edge_counts = np.asarray(edge_counts, dtype=np.int64)
    order = np.argsort(-edge_counts, kind="stable")
    shards = [[] for _ in range(world_size)]
    for i, gidx in enumerate(order):
        rnd, pos = divmod(i, world_size)
        r = pos if rnd % 2 == 0 else world_size - 1 - pos
        shards[r].append(int(gidx))
    return [np.asarray(sorted(s), dtype=np.int64) for s in shards]


class GfGather:
    """All-gather of the per-rank gf' rows into the full (G, DG') table in ORIGINAL graph order.

    Shards may differ by one graph, so every rank contributes `max_count` rows (zero padded) to
    all_gather_into_tensor; a precomputed index table undoes the padding and the partition permutation.
    On GPUs the collective runs on its own stream so that it overlaps the next forward (xGMI is point-to-point and
    the message is <= 64 KB/rank: the gather is latency-bound, so hiding it is what matters)."""

    def __init__(self, shards, rank, world_size, dg, device, group=None, overlap=True):
        self.shards, self.rank, self.world, self.dg, self.device, self.group = shards, rank, world_size, dg, device, group
        self.max_count = max(len(s) for s in shards)
        self.G = int(sum(len(s) for s in shards))
        src = np.zeros(self.G, dtype=np.int64)  # row of the gathered [world*max_count] table for original graph id
        for r, s in enumerate(shards):
            src[s] = r * self.max_count + np.arange(len(s))
        self.src_index = torch.from_numpy(src).to(device)
        self.send = torch.zeros((self.max_count, dg), dtype=torch.float32, device=device)
        self.recv = torch.empty((world_size * self.max_count, dg), dtype=torch.float32, device=device)
        self.is_cuda = torch.device(device).type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=device) if (self.is_cuda and overlap) else None
        self._ready = None
        self._ev = self._ready_ev = None

    def start(self, gf_local):
        """gf_local: (n_local, dg) rows of this rank's graphs (shard order).  Asynchronous on GPUs."""
        n = gf_local.shape[0]
        if self.comm_stream is None:
            self.send[:n].copy_(gf_local)
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group) if self.is_cuda else self._gather_cpu()
            return
        if self._ev is None:
            self._ev, self._ready_ev = torch.cuda.Event(), torch.cuda.Event()
        self._ev.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self._ev)
            self.send[:n].copy_(gf_local, non_blocking=True)
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
            self._ready_ev.record(self.comm_stream)
            self._ready = self._ready_ev

    def _gather_cpu(self):
        parts = [torch.empty_like(self.send) for _ in range(self.world)]
        dist.all_gather(parts, self.send, group=self.group)
        self.recv.copy_(torch.cat(parts, dim=0))

    def finish(self):
        """Full (G, dg) table in original graph order (waits for the collective on the current stream)."""
        if self._ready is not None:
            torch.cuda.current_stream(self.device).wait_event(self._ready)
            self._ready = None
        return self.recv.index_select(0, self.src_index)


def sharded_block_forward(forward_fn, x_local, gather: GfGather):
    """Runs `forward_fn` (a GNBlock / GNCore / model over this rank's shard) and all-gathers gf'.
    Returns (y_local, gf_all) with gf_all (G, DG') in original graph order on every rank.  `forward_fn` is the HIP
    path in production; the gloo CPU tests inject the oracle as the checker stand-in."""
    y = forward_fn(x_local)
    gf = y.gf if hasattr(y, "gf") else y["gf"]
    gf_rows = gf.permute(2, 1, 0)[0] if gf.dim() == 3 else gf  # Julia-shaped (DG, G_local, 1) → (G_local, DG)
    gather.start(gf_rows)
    return y, gather.finish()
