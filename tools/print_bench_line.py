"""Prints the figures of one bench.py JSON line (headline + secondary configs) in a few readable rows."""
import json
import sys

raw = open(sys.argv[1]).read().strip()
d = json.loads(raw) if raw.startswith("{\n") or raw.count("\n") > 3 else json.loads(raw.splitlines()[-1])  # a detail file (indented) or a one-line record
r = d["roofline"]
print("headline", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "whole", r.get("frac_whole_step"), "traffic", r.get("traffic"),
      "kernel_us", r.get("kernel_us"), "batch_ms", d.get("batch_ms"), "c_abi", d.get("c_abi_ms_per_step"),
      "pipelined", (d.get("pipelined_two_streams") or {}).get("ms_per_step"), "chained", (d.get("chained_graph_update") or {}).get("ms_per_step"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
for k, v in (d.get("secondary") or {}).items():
    if isinstance(v, dict):
        rr = v.get("roofline", {})
        print("  ", k, v.get("value"), "ms", v.get("ms_per_step"), "frac", rr.get("frac"), "whole", rr.get("frac_whole_step"), "traffic", rr.get("traffic"),
              "c_abi", v.get("c_abi_ms_per_step"), "cpu", (v.get("cpu_baseline") or {}).get("value"),
              "batch", {a: b for a, b in (v.get("batch_ms") or {}).items() if a != "what"}, "err", v.get("error"), "wall", v.get("wall_s"))
    else:
        print("  ", k, v)
print(len(json.dumps(d)), "bytes")
