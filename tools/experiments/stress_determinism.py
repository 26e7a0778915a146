#!/usr/bin/env python3
"""Repeated forwards of the core-dims block (the NL = 3 edge GEMM: LDS-DMA'd destination rows, untracked source-row loads behind a counted wait) and of a GNCore
(k_ffn_x6 for the edges, k_ffn_fused for the nodes) on C2 under load: every repetition must reproduce the first one bit for bit (a visibility / wait-count race would show up as a differing checksum)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import graphnets_jl_amd as gn

dev = torch.device("cuda", 0)
colptrs, rowvals, nn = bench.make_c2()
g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
din = dout = (128, 64, 32)
E, N, G = g.n_edges, g.n_nodes, g.n_graphs
mk = lambda T, d: torch.randn((1, T, d), device=dev).permute(2, 1, 0)
x = gn.NT(g, mk(E, din[0]), mk(N, din[1]), mk(G, din[2]))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for name, layer in (("GNBlock(128,64,32)", gn.GNBlock(din, dout, device=dev, act=("relu", "relu", "identity"))), ("GNCore(128,64,32)", gn.GNCore(din, device=dev))):
    ref = None
    bad = 0
    for i in range(reps):
        y = layer(x)
        chk = tuple(int(t.contiguous().view(torch.int32).to(torch.int64).sum().item()) for t in (y.ef, y.nf, y.gf))
        if ref is None:
            ref = chk
        elif chk != ref:
            bad += 1
            print(name, "repetition", i, "differs:", chk, "vs", ref, flush=True)
    print(f"{name}: {reps} repetitions, {bad} differing", flush=True)
    assert bad == 0
