#!/usr/bin/env python3
"""How long does the MI355X take to reach its sustained clock state under THIS load?  A captured forward is replayed back to back for `seconds`; every block
of replays is timed on the GPU (one pair of events per block).  what = c4 (Encoder -> 2 x GNCore(128,64,32) -> Decoder, ~4 ms per replay, blocks of 25),
block (the headline: GNBlock (10,5,0) => (3,4,5) on C2, ~25 us per replay, blocks of 4000) or core (GNBlock at core dims, ~0.46 ms, blocks of 200).
    python tools/experiments/clock_state_probe.py c4|block|core [seconds]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import graphnets_jl_amd as gn

what = sys.argv[1] if len(sys.argv) > 1 else "c4"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
dev = torch.device("cuda", 0)
colptrs, rowvals, nn = bench.make_c2()
g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
tg = torch.Generator(device=dev); tg.manual_seed(1)
mk = lambda T, d: torch.rand((1, T, d), generator=tg, device=dev).permute(2, 1, 0) if d else None
if what == "c4":
    model, _ = bench.c4_model(gn, torch, (128, 64, 32), dev)
    x = gn.NT(g, mk(g.n_edges, 10), mk(g.n_nodes, 5), None)
    per = 25
else:
    din, dout = ((10, 5, 0), (3, 4, 5)) if what == "block" else ((128, 64, 32), (128, 64, 32))
    model = [gn.GNBlock(din, dout, device=dev)]
    x = gn.NT(g, mk(g.n_edges, din[0]), mk(g.n_nodes, din[1]), mk(g.n_graphs, din[2]))
    per = 4000 if what == "block" else 200


def model_fn(t):
    y = t
    for layer in model:
        y = layer(y)
    return y


graphed = gn.Graphed(model_fn, x)
for _ in range(3):
    graphed.graph.replay()
torch.cuda.synchronize(dev)
time.sleep(1.0)  # start from an idle chip
print(f"{what}: replays back to back from idle, blocks of {per}: seconds since start -> ms per replay", flush=True)
t_start = time.perf_counter()
last = None
n = 0
while time.perf_counter() - t_start < seconds:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(per):
        graphed.graph.replay()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / per
    n += 1
    if last is None or abs(ms - last) > 0.004 * last or n % 20 == 0:  # print changes of state (> 0.4 %) and every 20th block
        print(f"  {time.perf_counter() - t_start:7.2f} s  {ms * (1000.0 if what == 'block' else 1.0):9.4f} {'us' if what == 'block' else 'ms'}", flush=True)
    last = ms
