#!/usr/bin/env python3
"""Does the platform keep in-stream order under many concurrent streams?  T host threads, each with its own torch stream, run a chain of small
order-dependent kernels (x <- x * a_i + b_i over a few KB .. MB) and compare with the serial result.  No gnx code involved.
python tools/experiments/stream_order_probe.py [T] [CHAIN] [ROUNDS] [N]"""
import sys
import threading
import torch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CHAIN = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 200
N = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 16
dev = torch.device("cuda", 0)
x0 = torch.rand(N, device=dev)
coef = [(1.0 + 0.001 * i, 0.01 * i) for i in range(CHAIN)]


def chain(x):
    y = x.clone()
    tmp = torch.empty_like(y)
    for a, b in coef:  # two dependent kernels per step through a scratch buffer (producer -> consumer, as a forward's intermediates)
        torch.mul(y, a, out=tmp)
        torch.add(tmp, b, out=y)
    return y


ref = chain(x0)
torch.cuda.synchronize()
bad = []


def worker(tid):
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for r in range(ROUNDS):
            y = chain(x0)
            st.synchronize()
            if not torch.equal(y, ref):
                bad.append((tid, r, int((y != ref).sum())))


ts = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
[t.start() for t in ts]
[t.join() for t in ts]
print(f"{T} threads x {ROUNDS} chains of {2 * CHAIN} dependent kernels over {N} floats: {len(bad)} wrong results {bad[:5]}")
