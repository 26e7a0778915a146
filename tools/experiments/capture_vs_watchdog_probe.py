"""Why bench.py and api.Graphed capture with capture_error_mode="thread_local": with a process group alive, torch's NCCL watchdog THREAD queries the
events of collectives in flight; a hipGraph capture in "global" mode (torch's default) makes such a call from another thread an error, thrown in the
watchdog thread -> std::terminate (seen once in ~20 starts of `bench.py --force-dist`, 2.5 s in, a backtrace through libstdc++'s thread trampoline).
  python tools/experiments/capture_vs_watchdog_probe.py global 300   /   ... thread_local 300        (fresh process each; prints ok or dies)"""
import os
import socket
import sys

import torch
import torch.distributed as dist

mode, n = sys.argv[1], int(sys.argv[2])
if "RANK" not in os.environ:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(1 << 20, device="cuda")
y = torch.empty_like(x)
for i in range(n):
    works = [dist.all_reduce(x, async_op=True) for _ in range(4)]  # collectives in flight: the watchdog polls their events
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg, capture_error_mode=mode):
        for _ in range(50):
            y = x * 2.0
    for w in works:
        w.wait()
    x.fill_(1.0)
torch.cuda.synchronize()
dist.destroy_process_group()
print("ok", mode, n)
