#!/usr/bin/env python3
"""N repetitions of one GNCore(128,64,32) forward and of the core-dims GNBlock on the C2 graph (100k nodes / 1M edges), every result compared bit for bit
with the first (the six-term kernels, k_node_x6, the one-launch core form, side streams).   python tools/experiments/core_determinism.py [N=100]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import graphnets_jl_amd as gn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
colptrs, rowvals, nn = bench.make_c2()
g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
d = (128, 64, 32)
mk = lambda T, w: torch.randn((1, T, w), device=dev).permute(2, 1, 0)
x = gn.NT(g, mk(g.n_edges, d[0]), mk(g.n_nodes, d[1]), mk(g.n_graphs, d[2]))
for name, layer in (("GNCore", gn.GNCore(d, device=dev).prepare()), ("GNBlock", gn.GNBlock(d, d, device=dev).prepare())):
    y0 = layer(x)
    ref = [t.clone() for t in (y0.ef, y0.nf, y0.gf)]
    bad = 0
    for i in range(N):
        y = layer(x)
        bad += sum(int(not torch.equal(a, b)) for a, b in zip((y.ef, y.nf, y.gf), ref))
    torch.cuda.synchronize()
    print(f"{name}{d} on C2: {N} repetitions, {bad} tensors differ from the first run")
