#!/usr/bin/env python3
"""Is the NARROW path (k_block_wave: packed FMAs, no matrix instruction) disturbed by a bf16 GEMM on another stream?  README-dims GNBlock on a 200k-edge
graph, 600 forwards compared with the serial result.   python tools/experiments/narrow_mix_probe.py [bf16|fp32|none]"""
import os
import sys
import threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import graphnets_jl_amd as gn
from tests import util as U

other = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rng = np.random.default_rng(7)
N, E = 20000, 200000
colptr, rowval = U.er_csc(rng, N, E)
g = gn.GNGraphBatch.from_csc([colptr], [rowval], [N])
din, dout = (10, 5, 3), (3, 4, 5)
blk = gn.GNBlock(din, dout)
x = U.to_nt(gn, g, *U.packed_inputs(rng, 1, E, N, 1, din))
ref = blk(x)
ref = tuple(t.clone() for t in (ref.ef, ref.nf, ref.gf))
torch.cuda.synchronize()
stop = threading.Event()
bad = []


def checker():
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(600):
            y = blk(x)
            st.synchronize()
            d = [n for n, a, b in zip(("ef", "nf", "gf"), (y.ef, y.nf, y.gf), ref) if not torch.equal(a, b)]
            if d:
                bad.append((i, d))
    stop.set()


def loader():
    if other == "none":
        return
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        X = torch.randn(2048, 2048, device="cuda").to(torch.bfloat16 if other == "bf16" else torch.float32)
        while not stop.is_set():
            X = (X @ X).clamp_(-1, 1)
            st.synchronize()


ts = [threading.Thread(target=checker), threading.Thread(target=loader)]
[t.start() for t in ts]
[t.join() for t in ts]
print(f"narrow GNBlock {din}=>{dout}, 600 forwards beside a {other} GEMM loop: {len(bad)} differ from the serial result {bad[:4]}")
