#!/usr/bin/env python3
"""gnx_row_stats (k_ln_stats_v4) alone, beside a torch bf16 GEMM loop on another stream: is the statistics kernel what goes wrong under concurrent load
(tests/overlap_probe.py: the LayerNorm-on-load forms fail, the materialised ones do not)?   python tests/stats_probe.py [rows=1200] [d=64] [iters=3000]"""
import ctypes as C
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import graphnets_jl_amd as gn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
lib = gn._lib.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
xs = [torch.rand((rows, d), device=dev) for _ in range(3)]


def stats(x, out):
    gn._lib.check(lib.gnx_row_stats(x.data_ptr(), rows, d, 1e-5, 0, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))


ref = []
for x in xs:
    o = torch.empty((rows, 2), device=dev)
    stats(x, o)
    ref.append(o)
torch.cuda.synchronize()
# the float64 answer, for what a wrong row looks like
stop = threading.Event()
bad = []


def victim():
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        out = torch.empty((rows, 2), device=dev)
        for it in range(iters):
            k = it % 3
            out.fill_(float("nan"))
            stats(xs[k], out)
            st.synchronize()
            if not torch.equal(out, ref[k]):
                wrong = torch.nonzero((out != ref[k]).any(dim=1)).flatten()
                if len(bad) < 6:
                    r0 = int(wrong[0])
                    bad.append({"it": it, "wrong_rows": int(wrong.numel()), "first_rows": wrong[:8].tolist(), "got": out[r0].tolist(), "ref": ref[k][r0].tolist(),
                                "nan_rows": int(torch.isnan(out).any(dim=1).sum())})
                else:
                    bad.append(it)
    stop.set()


def aggressor():
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        X = torch.randn(2048, 2048, device=dev).to(torch.bfloat16)
        while not stop.is_set():
            X = (X @ X).clamp_(-1, 1)
            st.synchronize()


ts = [threading.Thread(target=victim), threading.Thread(target=aggressor)]
[t.start() for t in ts]
[t.join() for t in ts]
print(json.dumps({"rows": rows, "d": d, "iters": iters, "wrong_runs": len(bad), "first": bad[:6]}))
