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


def partition_replicas(n_replicas, world_size):
    """A shared-adjacency batch (src/batch.jl:66: one graph, R = the data batch size) shards over its DATA BATCH the same way a vector of
    graphs shards by graph: the replicas are independent units — `batched_mul` never mixes batch indices.  Contiguous slices,
    equal counts (+-1): every rank keeps the one handle of the shared graph and its slice ef[:, :, r0:r1] / nf / gf.  Returns a list of
    int64 index arrays (replica ids, ascending) — a partition GfGather takes as it takes a partition of graphs."""
    R, W = int(n_replicas), int(world_size)
    assert R >= 0 and W >= 1
    cuts = [(R * r) // W for r in range(W + 1)]
    return [np.arange(cuts[r], cuts[r + 1], dtype=np.int64) for r in range(W)]


def sharded_replica_forward(forward_fn, x_local, gather: GfGather):
    """`sharded_block_forward` for a shared-adjacency batch sharded by replica (`partition_replicas`): runs `forward_fn` over this rank's
    replicas and all-gathers gf' — Julia-shaped (DG, 1, R_local) per rank — into (R, DG) in replica order on every rank."""
    y = forward_fn(x_local)
    gf = y.gf if hasattr(y, "gf") else y["gf"]
    gf_rows = gf.permute(2, 1, 0)[:, 0, :] if gf.dim() == 3 else gf  # (DG, 1, R_local) -> (R_local, DG)
    gather.start(gf_rows.contiguous())
    return y, gather.finish()


class DistBlockRunner:
    """ONE process driving n devices through the C boundary's own sharded path (gnx_dist_*: what the Julia shim's DistBlock binds) —
    the counterpart of GfGather for hosts that are not one-process-per-GPU.  Builds the communicator (ncclCommInitAll), one graph handle,
    one replicated parameter set and `n_sets` feature / output / workspace buffer sets per device; `run(first_set, n_steps)` is ONE
    gnx_dist_block_forward_steps: n_steps forwards per device (buffer sets first_set, first_set + 1, ... modulo n_sets), replayed as one
    hipGraph per device after the first call with the same sets, one grouped all-gather of the stacked gf' tables, and gf_all[r] =
    [n_steps][G][og] in ORIGINAL graph order on every device.

    shards: list of int64 arrays (original graph ids per rank, e.g. partition_graphs(...)); graphs_of(r) -> (colptrs, rowvals, n_nodes) of
    rank r's graphs in shard order; make_block(device) -> a GNBlock with the (replicated) parameters on that device."""

    def __init__(self, devices, shards, graphs_of, make_block, dims, n_sets=8, max_steps=1, seed=1234):
        import ctypes as C
        from . import _lib
        from .api import GNGraphBatch
        self.lib, self.C, self._lib = _lib.load(), C, _lib
        self.n = len(devices)
        self.devices = [torch.device("cuda", int(d)) for d in devices]
        (de, dn, dg), (oe, on, og) = dims
        self.dims, self.og, self.n_sets = dims, og, n_sets
        self.G = int(sum(len(s) for s in shards))
        off = np.zeros(self.n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in shards])
        ids = np.ascontiguousarray(np.concatenate([np.asarray(s, dtype=np.int64) for s in shards]))
        p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        self._d = C.c_void_p()
        devs = (C.c_int32 * self.n)(*[d.index for d in self.devices])
        _lib.check(self.lib.gnx_dist_create(devs, self.n, p64(off), p64(ids), self.G, og, C.byref(self._d)))
        self.keep, self.handles, self.params, self.sets, self.gall, self.streams, self.ws_bytes = [], [], [], [], [], [], []
        self.edges = 0
        for r, dev in enumerate(self.devices):
            with torch.cuda.device(dev):
                colptrs, rowvals, nn = graphs_of(r)
                g = GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
                blk = make_block(dev)
                bp = blk._c(self.keep)
                wsb = int(self.lib.gnx_block_workspace_bytes(g._h, C.byref(bp), 1))
                tg = torch.Generator(device=dev); tg.manual_seed(seed + r)
                mk = lambda T, d: torch.rand((T, d), generator=tg, device=dev, dtype=torch.float32) if d > 0 else None
                sets = [dict(ef=mk(g.n_edges, de), nf=mk(g.n_nodes, dn), gf=mk(g.n_graphs, dg),
                             eo=torch.empty((g.n_edges, oe), device=dev) if oe else None, no=torch.empty((g.n_nodes, on), device=dev) if on else None,
                             ws=torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)) for _ in range(n_sets)]
                self.handles.append(g); self.params.append(bp); self.keep.append(blk); self.sets.append(sets); self.ws_bytes.append(max(wsb, 256))
                self.gall.append(torch.empty((max_steps, self.G, og), device=dev))
                self.streams.append(torch.cuda.Stream(device=dev))
                self.edges += g.n_edges
        self.max_steps = max_steps

    def _ptrs(self, first_set, n_steps, key):
        C = self.C
        vals = []
        for s in range(n_steps):
            for r in range(self.n):
                t = self.sets[r][(first_set + s) % self.n_sets][key]
                vals.append(t.data_ptr() if t is not None else None)
        return (C.c_void_p * len(vals))(*vals)

    def run(self, first_set=0, n_steps=1, flags=0):
        C = self.C
        assert 1 <= n_steps <= self.max_steps
        per_rank = lambda f: (C.c_void_p * self.n)(*[f(r) for r in range(self.n)])
        nbytes = (C.c_size_t * self.n)(*self.ws_bytes)
        self._lib.check(self.lib.gnx_dist_block_forward_steps(
            self._d, n_steps, per_rank(lambda r: self.handles[r]._h.value), per_rank(lambda r: C.addressof(self.params[r])),
            self._ptrs(first_set, n_steps, "ef"), self._ptrs(first_set, n_steps, "nf"), self._ptrs(first_set, n_steps, "gf"),
            self._ptrs(first_set, n_steps, "eo"), self._ptrs(first_set, n_steps, "no"), per_rank(lambda r: self.gall[r].data_ptr()),
            self._ptrs(first_set, n_steps, "ws"), nbytes, flags, per_rank(lambda r: self.streams[r].cuda_stream)))

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        if self._d:
            self.synchronize()
            self._lib.check(self.lib.gnx_dist_destroy(self._d))
            self._d = self.C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
