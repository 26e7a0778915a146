#!/usr/bin/env python3
"""Config 4 at several sizes of the C2 graph (N nodes, 10 N edges): ms per replay of the captured forward (median of 5 regions of 20 replays behind 300 ms of load),
and the same per million edges.  Does the 4.000 ms of the 1M-edge forward scale with the work, or is it a floor?"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import graphnets_jl_amd as gn

dev = torch.device("cuda", 0)
model, _ = bench.c4_model(gn, torch, (128, 64, 32), dev)
for N in ([int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (100_000, 95_000, 90_000, 80_000, 60_000, 120_000, 100_000)):
    colptrs, rowvals, nn = bench.make_c2(N=N, E=10 * N)
    g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
    tg = torch.Generator(device=dev); tg.manual_seed(1)
    x = gn.NT(g, torch.rand((1, g.n_edges, 10), generator=tg, device=dev).permute(2, 1, 0), torch.rand((1, g.n_nodes, 5), generator=tg, device=dev).permute(2, 1, 0), None)

    def model_fn(t):
        y = t
        for layer in model:
            y = layer(y)
        return y

    graphed = gn.Graphed(model_fn, x)
    bench.spin_up(torch, dev, graphed.graph.replay, 300.0)
    regs = []
    for _ in range(5):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(20):
            graphed.graph.replay()
        torch.cuda.synchronize(dev)
        regs.append((time.perf_counter() - t0) * 1e3 / 20)
    ms = float(np.median(regs))
    print(f"  N = {N:7d}, E = {g.n_edges:8d}: {ms:.4f} ms per forward = {ms / (g.n_edges / 1e6):.4f} ms per million edges   (regions {' '.join(f'{v:.3f}' for v in regs)})", flush=True)
    del graphed, x, g
