#!/usr/bin/env python3
"""Per-wave timeline of k_block_wave on C2 (diagnostic build: tools/build_variant.sh wstamps gnx_narrow.hip -DGNX_WAVE_STAMPS_BUILD).
  GNX_LIB_PATH=graphnets.jl_amd/libgnx_wstamps.so python3 tools/wave_stamps.py
Stamps per wave (shader clocks): 0 start, 1 tile record arrived, 2 every load issued, 3 node-side preparation done (colptr + own nf
row arrived), 4 edge phase done (ef rows, rowval, gathered rows arrived; stores issued), 5 node phase done, 6 end."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = os.path.abspath("gpurun_out/wave_stamps.bin")
os.makedirs(os.path.dirname(dump), exist_ok=True)
import torch
import bench
import graphnets_jl_amd as gn

dev = torch.device("cuda", 0)
colptrs, rowvals, nn = bench.make_c2()
g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
din, dout = bench.DIMS["readme"]
blk = gn.GNBlock(din, dout, device=dev)
E, N, G = g.n_edges, g.n_nodes, g.n_graphs
mk = lambda T, d: torch.rand((1, T, d), device=dev) if d > 0 else None
flush = torch.empty(512 * 1024 * 1024 // 4, device=dev)
xs = [gn.NT(g, *(None if t is None else t.permute(2, 1, 0) for t in (mk(E, din[0]), mk(N, din[1]), mk(G, din[2])))) for _ in range(3)]
for x in xs:
    blk(x)
torch.cuda.synchronize()
os.environ["GNX_WAVE_STAMPS_DUMP"] = dump
res = []
for rep in range(4):
    flush.zero_()  # cache-cold
    torch.cuda.synchronize()
    blk(xs[rep % 3])
    torch.cuda.synchronize()
    a = np.fromfile(dump, dtype=np.uint64).reshape(-1, 8)
    a = a[a[:, 6] > 0]
    t = a[:, :7].astype(np.int64)
    xcc = (a[:, 7] >> np.uint64(32)).astype(np.int64) & 0xF
    t0 = np.zeros(len(t), np.int64)
    for x in np.unique(xcc):  # one clock domain per XCC: relative to the first wave start of that XCC
        t0[xcc == x] = t[xcc == x, 0].min()
    rel = t - t0[:, None]
    ph = np.diff(t, axis=1)
    q = lambda v: "p10 %6d  p50 %6d  p90 %6d" % tuple(np.percentile(v, [10, 50, 90]))
    print(f"rep {rep}: {len(t)} waves; span (last end - first start, per XCC) max {rel[:, 6].max()} clocks")
    print("  wave start after kernel start:", q(rel[:, 0]))
    for i, name in enumerate(["tile record", "issue all loads", "wait colptr/nf + node prep", "edge phase (wait ef/rowval/gather, compute, stores)", "node phase", "partial sums"]):
        print(f"  {name:55s}", q(ph[:, i]))
    print("  wave lifetime:", q(t[:, 6] - t[:, 0]), "  wave end after kernel start:", q(rel[:, 6]))
