#!/usr/bin/env python3
"""Times forward + backward of ONE GNCore(128,64,32) (and of a GNBlock at core dims) on the C2 graph (100k nodes / 1M edges), per-kernel scopes included.
The backward's entry points carry no flags: run once as is (six bf16 terms) and once with GNX_FFN_FP32=1 GNX_EDGE_FP32=1 (the fp32 matrix instruction).
python tools/experiments/bw_time.py [scale=1.0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import graphnets_jl_amd as gn  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda", 0)
colptrs, rowvals, nn = bench.make_c2(seed=2, N=int(100_000 * scale), E=int(1_000_000 * scale))
g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn, device=dev)
dims = (128, 64, 32)
core = gn.GNCore(dims, device=dev)
for t in core.parameters():
    t.requires_grad_(True)
tg = torch.Generator(device=dev); tg.manual_seed(1)
mk = lambda T, d: torch.rand((1, T, d), generator=tg, device=dev).permute(2, 1, 0).requires_grad_(True)
x = gn.NT(g, mk(g.n_edges, 128), mk(g.n_nodes, 64), mk(g.n_graphs, 32))


def step():
    y = core(x)
    loss = y.ef.sum() + y.nf.sum() + y.gf.sum()
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
gn.profile_reset(); gn.profile_enable(True)
for _ in range(3):
    step()
torch.cuda.synchronize()
gn.profile_enable(False)
prof = gn.profile_read()
print("forms:", "fp32 instruction" if os.environ.get("GNX_EDGE_FP32") else "six bf16 terms", " fwd+bwd of one GNCore(128,64,32) at %d edges: %.3f ms" % (g.n_edges, dt * 1e3))
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    print("   %-24s %8.1f us per step (%d launches)" % (k, v["total_ms"] * 1e3 / 3, v["launches"] // 3))
