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
    deal in a snake/boustrophedon order) — `gnx_dist_partition` of the C boundary (one implementation for the Python
    mirror, the Julia shim and C hosts).  Returns a list of int64 index arrays (original graph ids, ascending)."""
    import ctypes as C
    from . import _lib
    counts = np.ascontiguousarray(edge_counts, dtype=np.int64)
    off = np.zeros(world_size + 1, dtype=np.int64)
    ids = np.zeros(len(counts), dtype=np.int64)
    p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    _lib.check(_lib.load().gnx_dist_partition(p64(counts), len(counts), int(world_size), p64(off), p64(ids)))
    return [ids[off[r]:off[r + 1]].copy() for r in range(world_size)]


def gather_plan(shards):
    """`gnx_dist_gather_plan` (host only): validates the partition and returns (src_row[G] int32, max_count) — row of original graph g
    in the gathered [world * max_count] table.  The plan gnx_dist_create builds, and the one GfGather indexes with."""
    import ctypes as C
    from . import _lib
    off = np.zeros(len(shards) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in shards])
    ids = np.ascontiguousarray(np.concatenate([np.asarray(s, dtype=np.int64) for s in shards]) if len(shards) else np.zeros(0, np.int64))
    G = int(off[-1])
    src = np.zeros(G, dtype=np.int32)
    mc = C.c_int64(0)
    p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    _lib.check(_lib.load().gnx_dist_gather_plan(p64(off), p64(ids), len(shards), G, src.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(mc)))
    return src, int(mc.value)


class GfGather:
    """All-gather of the per-rank gf' rows into the full (G, DG') table in ORIGINAL graph order.

    Shards may differ by one graph, so every rank contributes `max_count` rows (zero padded) to
    all_gather_into_tensor; a precomputed index table undoes the padding and the partition permutation.
    `stack` = M > 1: M tables (the gf' of M consecutive steps) travel in ONE collective — send is [M][max_count][dg],
    the result [M][G][dg] ("fewer, larger collectives": one table is <= 64 KB, pure latency on xGMI).
    On GPUs the collective runs on its own stream so that it overlaps the next forward (xGMI is point-to-point and
    the message is small: the gather is latency-bound, so hiding it is what matters)."""

    def __init__(self, shards, rank, world_size, dg, device, group=None, overlap=True, stack=1):
        self.shards, self.rank, self.world, self.dg, self.device, self.group = shards, rank, world_size, dg, device, group
        self.stack = int(stack)
        plan, self.max_count = gather_plan(shards)  # the C boundary's plan (one implementation): row r * max_count + k
        self.G = int(sum(len(s) for s in shards))
        M, mc = self.stack, self.max_count
        # row of the gathered [world][M][max_count] table for (step m, original graph id): rank r's M stacked tables are consecutive
        rank_of, k_of = plan.astype(np.int64) // max(mc, 1), plan.astype(np.int64) % max(mc, 1)
        src = np.stack([(rank_of * M + m) * mc + k_of for m in range(M)])
        self.src_index = torch.from_numpy(src.reshape(-1)).to(device)
        self.send = torch.zeros((M, mc, dg), dtype=torch.float32, device=device)
        self.recv = torch.empty((world_size * M * mc, dg), dtype=torch.float32, device=device)
        self.is_cuda = torch.device(device).type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=device) if (self.is_cuda and overlap) else None
        self._ready = None
        self._ev = self._ready_ev = None

    def start(self, gf_local):
        """gf_local: (n_local, dg) rows of this rank's graphs in shard order — or (M, n_local, dg) with stack = M.
        Asynchronous on GPUs."""
        g3 = gf_local if gf_local.dim() == 3 else gf_local[None]
        n = g3.shape[1]
        if self.comm_stream is None:
            self.send[:, :n].copy_(g3)
            self.start_inplace()
            return
        if self._ev is None:
            self._ev, self._ready_ev = torch.cuda.Event(), torch.cuda.Event()
        self._ev.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self._ev)
            self.send[:, :n].copy_(g3, non_blocking=True)
            dist.all_gather_into_tensor(self.recv, self.send.view(-1, self.dg), group=self.group)
            self._ready_ev.record(self.comm_stream)
            self._ready = self._ready_ev

    def start_inplace(self):
        """The forward wrote its gf' rows straight into `self.send[m, :n_local]` (no staging copy)."""
        if self.comm_stream is None:
            if self.is_cuda:
                dist.all_gather_into_tensor(self.recv, self.send.view(-1, self.dg), group=self.group)
            else:
                self._gather_cpu()
            return
        if self._ev is None:
            self._ev, self._ready_ev = torch.cuda.Event(), torch.cuda.Event()
        self._ev.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self._ev)
            dist.all_gather_into_tensor(self.recv, self.send.view(-1, self.dg), group=self.group)
            self._ready_ev.record(self.comm_stream)
            self._ready = self._ready_ev

    def _gather_cpu(self):
        flat = self.send.view(-1, self.dg)
        parts = [torch.empty_like(flat) for _ in range(self.world)]
        dist.all_gather(parts, flat, group=self.group)
        self.recv.copy_(torch.cat(parts, dim=0))

    def result(self, out=None):
        """(G, dg) — or (M, G, dg) with stack = M — in original graph order, from the last completed gather.
        `out`: a preallocated (M*G, dg) tensor to write into (no allocation on the hot path)."""
        if out is not None:
            torch.index_select(self.recv, 0, self.src_index, out=out)
        else:
            out = self.recv.index_select(0, self.src_index)
        return out.view(self.stack, self.G, self.dg) if self.stack > 1 else out.view(self.G, self.dg)

    def finish(self, out=None):
        """Waits for the collective on the current stream, then `result()`."""
        if self._ready is not None:
            torch.cuda.current_stream(self.device).wait_event(self._ready)
            self._ready = None
        return self.result(out)


def sharded_block_forward(forward_fn, x_local, gather: GfGather):
    """Runs `forward_fn` (a GNBlock / GNCore / model over this rank's shard) and all-gathers gf'.
    Returns (y_local, gf_all) with gf_all (G, DG') in original graph order on every rank.  `forward_fn` is the HIP
    path in production; the gloo CPU tests inject the oracle as the checker stand-in."""
    y = forward_fn(x_local)
    gf = y.gf if hasattr(y, "gf") else y["gf"]
    gf_rows = gf.permute(2, 1, 0)[0] if gf.dim() == 3 else gf  # Julia-shaped (DG, G_local, 1) → (G_local, DG)
    gather.start(gf_rows)
    return y, gather.finish()
