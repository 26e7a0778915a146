"""How much would co-scheduling Z independent 1M-edge steps in ONE launch buy?  The replica dimension of the kernels already does exactly that
for a shared graph (blockIdx.y = replica: R feature sets through one graph structure), so: C2 at README dims with R = 1, 2, 4, 8 replicas per
launch, one hipGraph of 20 launches over rotating buffer sets (cache-cold), per launch and per replica.   python tools/experiments/replica_fusion_probe.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
import graphnets_jl_amd as gn

colptrs, rowvals, nn = bench.make_c2(seed=2, N=100_000, E=1_000_000)
g = gn.GNGraphBatch.from_csc(colptrs, rowvals, nn)
din, dout = (10, 5, 0), (3, 4, 5)
blk = gn.GNBlock(din, dout)
dev = g.device
for R in (1, 2, 4, 8):
    plan = gn.BlockPlan(blk, g, R=R)
    nsets = max(2, 16 // R)
    mk = lambda T, d: torch.rand((R, T, d), device=dev) if d > 0 else None
    sets = [dict(ef=mk(g.n_edges, din[0]), nf=mk(g.n_nodes, din[1]), gf=None, out=plan.outputs(), ws=plan.new_workspace()) for _ in range(nsets)]
    K = 20
    for form in ("two-launch", "steps"):
        def run():
            if form == "steps":
                plan.steps([sets[i % nsets] for i in range(K)])
            else:
                for i in range(K):
                    b = sets[i % nsets]
                    plan(b["ef"], b["nf"], b["gf"], *b["out"], ws=b["ws"])
        run(); torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg, capture_error_mode="thread_local"):
            run()
        for _ in range(30):
            cg.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); cg.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[2] / K
        print(f"R={R} {form:10s}: {dt * 1e6:7.2f} us per launch-step, {dt * 1e6 / R:6.2f} us per replica, {R * 1e6 / dt / 1e9:6.1f} G edges/s, "
              f"{60_000_580 * R / dt / 8e12:.3f} of the roof on the whole step", flush=True)
