"""Start-up of a one-rank RCCL process group as bench.py --force-dist makes it, N times in fresh processes: how often does it die, and with what?
  python tools/experiments/rccl_init_probe.py 30     (prints one line per failed start with the head and the tail of its stderr)"""
import os
import socket
import subprocess
import sys

CHILD = r'''
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(8, device="cuda"); out = torch.empty(8, device="cuda")
dist.all_gather_into_tensor(out, x); torch.cuda.synchronize()
dist.barrier(); dist.destroy_process_group()
print("ok")
'''


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


n, bad = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 0
for i in range(n):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=300)
    if r.returncode != 0 or "ok" not in r.stdout:
        bad += 1
        e = r.stderr.strip()
        print(f"start {i}: exit code {r.returncode}\n--- head\n{e[:1500]}\n--- tail\n{e[-800:]}", flush=True)
print(f"{bad} of {n} starts failed")
