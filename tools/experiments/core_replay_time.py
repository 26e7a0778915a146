#!/usr/bin/env python3
"""ms per replay of ONE captured GNCore(128,64,32) forward on C2 (median of 7 regions of 40 replays behind 300 ms of load) — an A/B target for switches that
the locked config-4 model time cannot resolve.   python tools/experiments/core_replay_time.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import graphnets_jl_amd as gn

dev = torch.device("cuda", 0)
colptrs, rowvals, nn = bench.make_c2()
g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
d = (128, 64, 32)
mk = lambda T, w: torch.randn((1, T, w), device=dev).permute(2, 1, 0)
x = gn.NT(g, mk(g.n_edges, d[0]), mk(g.n_nodes, d[1]), mk(g.n_graphs, d[2]))
core = gn.GNCore(d, device=dev)
if "--no-prepare" not in sys.argv:
    core.prepare()  # (the weight planes made once, as a model's layers have them)
graphed = gn.Graphed(lambda t: core(t), x)
bench.spin_up(torch, dev, graphed.graph.replay, 300.0)
regs = []
for _ in range(7):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(40):
        graphed.graph.replay()
    torch.cuda.synchronize(dev)
    regs.append((time.perf_counter() - t0) * 1e3 / 40)
print(f"{np.median(regs):.4f} ms per GNCore forward   (regions {' '.join(f'{v:.3f}' for v in regs)})", flush=True)
