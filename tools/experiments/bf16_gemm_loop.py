#!/usr/bin/env python3
"""A bf16 (or fp32) GEMM loop for N seconds: the foreign matrix load of the mfma-mix experiments.  python tools/experiments/bf16_gemm_loop.py [seconds] [bf16|fp32]"""
import sys
import time
import torch
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
X = torch.randn(2048, 2048, device="cuda").to(dt)
torch.cuda.synchronize(); print("started", flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20):
        X = (X @ X).clamp_(-1, 1)
    torch.cuda.synchronize(); n += 20
print(f"{n} GEMMs of 2048^3 in {dt}")
