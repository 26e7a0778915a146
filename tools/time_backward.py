#!/usr/bin/env python3
"""Forward+backward time of one GNBlock (and one GNCore) on the C2 graph through torch autograd, with the per-kernel
split of the backward.   python tools/time_backward.py [de,dn,dg:oe,on,og ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import graphnets_jl_amd as gn  # noqa: E402

cp, rv, nn = bench.make_c2()
g = gn.GNGraphBatch.from_csc(cp, rv, nn)
dev = g.device


def run(din, dout, core=False):
    layer = gn.GNCore(din) if core else gn.GNBlock(din, dout)
    ps = layer.parameters() if core else [t for l in (layer.edgefn, layer.nodefn, layer.graphfn) for t in (l.weight, l.bias)]
    for p in ps:
        p.requires_grad_(True)
    mk = lambda rows, d: torch.rand((1, rows, d), device=dev, requires_grad=True) if d else None
    ef, nf, gf = mk(g.n_edges, din[0]), mk(g.n_nodes, din[1]), mk(1, din[2])
    jl = lambda t: None if t is None else t.permute(2, 1, 0)
    x = gn.NT(g, jl(ef), jl(nf), jl(gf))

    def it():
        y = layer(x)
        sum(t.sum() for t in (y.ef, y.nf, y.gf) if t is not None).backward()

    for _ in range(2):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        it()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    gn._lib.profile_enable(True); gn._lib.profile_reset()
    it(); torch.cuda.synchronize()
    prof = {k: round(v["total_ms"] * 1e3, 1) for k, v in gn._lib.profile_read().items() if not k.startswith("__")}
    gn._lib.profile_enable(False)
    print(f"{'GNCore' if core else 'GNBlock'} {din}=>{dout}: forward+backward {dt * 1e3:.3f} ms / iteration; kernels (us): {prof}")


specs = sys.argv[1:] or ["10,5,0:3,4,5", "128,64,32:128,64,32"]
for sp in specs:
    core = sp.startswith("core:")
    a, b = sp.replace("core:", "").split(":")
    run(tuple(int(v) for v in a.split(",")), tuple(int(v) for v in b.split(",")), core)
