#!/usr/bin/env python3
"""Is the config-4 replay loop paced by the HOST?  K back-to-back replays of the captured forward + one synchronisation, for several K: ms per replay, and the
host time every single graph.replay() call takes.  (If the submission path blocks on a timer once some queue is full, the per-call host time jumps to the
timer's period and the loop reads that period per step whatever the kernels do.)
    python tools/experiments/replay_pacing_probe.py"""
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
model, _ = bench.c4_model(gn, torch, (128, 64, 32), dev)
tg = torch.Generator(device=dev); tg.manual_seed(1)
x = gn.NT(g, torch.rand((1, g.n_edges, 10), generator=tg, device=dev).permute(2, 1, 0), torch.rand((1, g.n_nodes, 5), generator=tg, device=dev).permute(2, 1, 0), None)


def model_fn(t):
    y = t
    for layer in model:
        y = layer(y)
    return y


graphed = gn.Graphed(model_fn, x)
bench.spin_up(torch, dev, graphed.graph.replay, 300.0)
print("K replays + one synchronisation: ms per replay (median of 5 regions) | host ms spent inside each replay() call of the last region", flush=True)
for K in (1, 2, 3, 4, 6, 8, 12, 20, 40, 100):
    regs = []
    for rep in range(5):
        bench.spin_up(torch, dev, graphed.graph.replay, 50.0)
        torch.cuda.synchronize(dev)
        calls = []
        t0 = time.perf_counter()
        for _ in range(K):
            c0 = time.perf_counter()
            graphed.graph.replay()
            calls.append((time.perf_counter() - c0) * 1e3)
        torch.cuda.synchronize(dev)
        regs.append((time.perf_counter() - t0) * 1e3 / K)
    cs = np.array(calls)
    print(f"  K = {K:4d}: {np.median(regs):.4f} ms per replay | replay() calls: first {cs[0]:.3f}, median {np.median(cs):.3f}, max {cs.max():.3f} ms; "
          + " ".join(f"{c:.2f}" for c in cs[:12]), flush=True)
