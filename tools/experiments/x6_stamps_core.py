#!/usr/bin/env python3
"""Shader-clock stamps of the six-term kernels inside ONE eager GNCore(128,64,32) forward on C2 (diagnostic build: tools/build_variant.sh x6st gnx_ffn_x6.hip
-DGNX_X6_STAMPS_BUILD after `git apply tools/experiments/x6_edge_form_stamps.patch`; run with GNX_LIB_PATH=graphnets.jl_amd/libgnx_x6st.so GNX_X6_STAMPS=1).
The launcher prints the per-workgroup phase averages on stderr; GNX_CORE_EDGE_SPLIT=1 shows the two-launch form's FeedForward beside it."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    y = core(x)
torch.cuda.synchronize(dev)
