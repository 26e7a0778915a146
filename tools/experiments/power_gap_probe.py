#!/usr/bin/env python3
"""Is config 4 bound by the socket's power management?  The captured forward (Encoder -> 2 x GNCore(128,64,32) -> Decoder on C2) is replayed with an idle
gap of g milliseconds after every replay; each replay is timed on the GPU with its own pair of events.  If the forward gets faster as the duty cycle
drops, its kernels run at a clock that the power limit sets, not at the clock the chip can reach.
    python tools/experiments/power_gap_probe.py [replays per gap]"""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bench.spin_up(torch, dev, graphed.graph.replay, 300.0)
print("idle gap after every replay (ms) -> GPU time of one forward (ms): median, min, max over", reps, "replays (the first 10 of a series dropped)", flush=True)
if len(sys.argv) > 2:  # a long gap-free series in blocks of 50: does the forward drift under sustained load?
    for blk in range(int(sys.argv[2])):
        ts = []
        for i in range(50):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); graphed.graph.replay(); e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize(dev)
        ms = np.array([a.elapsed_time(b) for a, b in ts])
        print(f"  replays {50 * blk:5d}..{50 * blk + 49:5d}: median {np.median(ms):.3f}  min {ms.min():.3f}  max {ms.max():.3f}", flush=True)
for gap in (0.0, 1.0, 2.0, 4.0, 8.0, 16.0, 0.0):
    ts = []
    for i in range(reps + 10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graphed.graph.replay(); e1.record()
        if gap > 0:
            e1.synchronize()
            time.sleep(gap * 1e-3)
        ts.append((e0, e1))
    torch.cuda.synchronize(dev)
    ms = np.array([a.elapsed_time(b) for a, b in ts[10:]])
    print(f"  gap {gap:5.1f} ms: {np.median(ms):.3f}  {ms.min():.3f}  {ms.max():.3f}   duty cycle {np.median(ms) / (np.median(ms) + gap):.2f}", flush=True)
