#!/usr/bin/env python3
"""Driver of tools/experiments/lds_hazard_repro.hip: the cut-down LayerNorm-on-load site in a loop on one stream, a torch bf16 GEMM loop (hipBLASLt)
on another; counts the (round, lane) pairs whose statistics were not what the workgroup had just written, per lane of the wave.
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o /tmp/lds_hazard_repro.so tools/experiments/lds_hazard_repro.hip
    python tools/experiments/lds_hazard_repro.py [guard=0|1] [launches=400] [blocks=75] [rounds=64] [aggressor=bf16|none]"""
import ctypes as C
import json
import sys
import threading

import torch

guard = int(sys.argv[1]) if len(sys.argv) > 1 else 0
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 400
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 75
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 64
aggr = sys.argv[5] if len(sys.argv) > 5 else "bf16"
lib = C.CDLL("/tmp/lds_hazard_repro.so")
lib.lds_hazard_victim.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
x = torch.rand(4096 * 4, device=dev)
bad = torch.zeros(64, dtype=torch.int64, device=dev)
sink = torch.zeros(4, device=dev)
stop = threading.Event()


def victim():
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(launches):
            rc = lib.lds_hazard_victim(x.data_ptr(), rounds, blocks, guard, bad.data_ptr(), sink.data_ptr(), st.cuda_stream)
            assert rc == 0, rc
            st.synchronize()
    stop.set()


def aggressor():
    if aggr == "none":
        return
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
torch.cuda.synchronize()
b = bad.cpu().tolist()
print(json.dumps({"guard": guard, "aggressor": aggr, "launches": launches, "blocks": blocks, "rounds": rounds, "checks": launches * blocks * rounds * 512 * 2,
                  "bad_total": sum(b), "bad_by_lane_quarter": [sum(b[0:16]), sum(b[16:32]), sum(b[32:48]), sum(b[48:64])], "bad_lanes": [i for i, v in enumerate(b) if v]}))
