#!/usr/bin/env python3
"""Platform check, no gnx code: does a torch fp32 GEMM (rocBLAS / hipBLASLt) stay bit-reproducible while a bf16 GEMM runs on another stream?
Thread 0 repeats C = A @ B in fp32 and compares with its own serial result; thread 1 loops a GEMM in the given dtype.
python tools/experiments/mfma_mix_probe.py [other_dtype=bf16|fp16|fp32|none] [n=512] [iters=2000]"""
import sys
import threading
import torch

other = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
A = torch.randn(n, n, device=dev, generator=g)
B = torch.randn(n, n, device=dev, generator=g)
ref = A @ B
torch.cuda.synchronize()
stop = threading.Event()
bad = []


def checker():
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(iters):
            C = A @ B
            st.synchronize()
            if not torch.equal(C, ref):
                d = (C - ref).abs()
                bad.append((i, int((d > 0).sum()), float(d.max())))
    stop.set()


def loader():
    if other == "none":
        return
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[other]
    with torch.cuda.stream(st):
        X = torch.randn(2048, 2048, device=dev).to(dt)
        while not stop.is_set():
            X = (X @ X).clamp_(-1, 1)
            st.synchronize()


ts = [threading.Thread(target=checker), threading.Thread(target=loader)]
[t.start() for t in ts]
[t.join() for t in ts]
print(f"fp32 GEMM {n}x{n} x {iters} beside a {other} GEMM loop on another stream: {len(bad)} results differ from the serial one {bad[:5]}")
