"""Compares the per-kernel durations bench.py reports (gnx_profile_*: dispatch timestamps) with rocprofv3's kernel trace of the same process.
The per-kernel pass of bench.py is the LAST K eager steps before the calibration launches (gnx::k_null): the trace rows between the last
hipGraph replay and the first k_null are exactly the launches the library timed."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
line = None
for l in open(os.path.join(out, "kt.log")):
    l = l.strip()
    if l.startswith("{") and '"roofline"' in l:
        line = json.loads(l)
assert line is not None, "no bench line in kt.log"
roof = line["roofline"]
K = line["steps"]
rows = []
for f in glob.glob(os.path.join(out, "kt", "**", "*_kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
first_null = next(i for i, n in enumerate(names) if "k_null" in n)
# walk back from the first k_null over the K steps of the per-kernel pass (skipping torch's spin kernel)
pass_rows = [r for r in rows[:first_null] if "gnx::" in r["Kernel_Name"]]
per_step = None
for n in range(1, 64):  # kernels per step: the smallest period of the tail
    tail = [r["Kernel_Name"] for r in pass_rows[-n * K:]]
    if len(tail) == n * K and all(tail[i] == tail[i % n] for i in range(n * K)) and len(set(tail[:n])) >= 1:
        per_step = n
        break
assert per_step, "could not find the per-kernel pass in the trace"
sel = pass_rows[-per_step * K:]
prof = defaultdict(list)
for r in sel:
    prof[r["Kernel_Name"].split("(")[0].replace("gnx::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"# {line['config']['workload'][:60]} dims {line['config']['dims']}: {K} steps, {per_step} kernels per step")
print("# library (gnx_profile_*, per scope):", json.dumps(roof.get("all_kernels_us")), "null", roof.get("null_kernel_us"))
tot = 0.0
for k, v in prof.items():
    print(f"rocprofv3  {k[:70]:70s} n={len(v):4d} avg {sum(v) / len(v):9.3f} us")
    tot += sum(v) / K
null = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_null" in r["Kernel_Name"]]
print(f"rocprofv3  k_null avg {sum(null) / len(null):.3f} us; sum of kernels per step {tot:.3f} us; library sum {sum(roof['all_kernels_us'].values()):.3f} us; "
      f"ratio {sum(roof['all_kernels_us'].values()) / tot:.4f}")
