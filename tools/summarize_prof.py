#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (gpurun_out/prof_<tag>/...) into small, committable files under profiles/.

  python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name> [--dims readme]
writes <name>_kernel_stats.csv (gnx kernels: calls, avg/min/max ns) and <name>_pmc.json (per-kernel mean of every
collected counter; FETCH_SIZE / WRITE_SIZE converted to bytes with the gfx950 corrections of MI355X_MICROARCH.md §HBM:
both counters are in KiB; FETCH_SIZE under-reports wide coalesced reads by exactly 2x, so it is doubled)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


FULL_NAMES = "--full-names" in sys.argv  # keep the template arguments of every kernel (one row per instantiation)


def short(name):
    """gnx::k_rows_gemm<128, true>(gnx::WideArgs) -> k_rows_gemm<128,true>"""
    name = name.split("(")[0].split("::")[-1].replace(" ", "")
    return name if (FULL_NAMES or name.startswith("k_rows_gemm") or name.startswith("k_ffn_fused") or name.startswith("k_ffn_x6<")) else name.split("<")[0]  # (k_ffn_fused / k_ffn_x6 <128> and <64> are different kernels: never averaged together)


# at core dims (template arguments: BN, quad outputs, K chunk, epilogue operand streams (3: destination rows through LDS), transcendental activation, loader class)
ALIASES = {"k_rows_gemm<128,true,32,2,false,0>": "k_rows_gemm_edge", "k_rows_gemm<128,true,32,3,false,0>": "k_rows_gemm_edge", "k_edge_x6": "k_rows_gemm_edge", "k_ffn_x6<128,false,true>": "k_core_edge_x6", "k_ffn_x6<128,true,true>": "k_core_edge_x6", "k_rows_gemm<64,true,32,0,false,1>": "k_rows_gemm_node",
           "k_rows_gemm<128,true,32,0,false,0>": "k_rows_gemm_proj"}


def main():
    src, dst = sys.argv[1], sys.argv[2]
    dims = sys.argv[sys.argv.index("--dims") + 1] if "--dims" in sys.argv else None
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    stats = defaultdict(list)
    for f in glob.glob(os.path.join(src, "kt", "**", "*_kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "gnx::" in row["Kernel_Name"]:
                stats[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    with open(dst + "_kernel_stats.csv", "w") as out:
        out.write("kernel,calls,avg_ns,min_ns,max_ns,total_ns,median_ns\n")  # (median: what the kernel takes at settled clocks; the average includes the post-idle transients)
        for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
            out.write(f"{k},{len(v)},{sum(v) / len(v):.1f},{min(v)},{max(v)},{sum(v)},{sorted(v)[len(v) // 2]}\n")
    pmc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "gnx::" in row["Kernel_Name"]:
                pmc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, cs in pmc.items():
        res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
        res[k]["launches_sampled"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
            fetch = res[k].get("FETCH_SIZE", 0.0) * 1024 * 2  # KiB -> B, x2 gfx950 wide-read correction
            write = res[k].get("WRITE_SIZE", 0.0) * 1024
            res[k]["hbm_fetch_bytes_corrected"] = fetch
            res[k]["hbm_write_bytes"] = write
            res[k]["hbm_bytes_per_launch"] = fetch + write
        if k in stats:
            res[k]["avg_ns"] = sum(stats[k]) / len(stats[k])
    with open(dst + "_pmc.json", "w") as out:
        json.dump(res, out, indent=1, sort_keys=True)
    if dims:
        # provenance: bench.py quotes these numbers only while the kernel sources they were measured on are the ones running
        import datetime
        import subprocess
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        try:
            commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
        except Exception:
            commit = None
        commit = os.environ.get("GNX_PROFILE_COMMIT", commit)
        with open(os.path.join(os.path.dirname(dst) or ".", f"traffic_{dims}.json"), "w") as out:
            tr = {k: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"]} for k, v in res.items() if "hbm_bytes_per_launch" in v}
            tr.update({ALIASES[k]: v for k, v in list(tr.items()) if k in ALIASES})
            din, dout = bench.DIMS[dims.split("_")[0]]
            tr["_meta"] = {"source_sha": bench.kernel_source_sha(din, dout), "commit": commit, "date": datetime.date.today().isoformat(),
                           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; KiB -> B, FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH §HBM)"}
            json.dump(tr, out, indent=1)
    if "--model-traffic" in sys.argv:
        # whole-model traffic of `bench.py --model c4`: every gnx kernel's bytes over the profiled run / the forwards the run executed
        # (bench.py prints "forwards_executed"; Graphed's two warm-up calls included) -> profiles/traffic_<key>.json, entry "__model__"
        import datetime
        import re
        import subprocess
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        key = sys.argv[sys.argv.index("--model-traffic") + 1]
        core = (128, 64, 32) if key == "c4" else tuple(int(v) for v in key.split("_")[1].split("-"))
        # bench.py's warm-up loops are time-based, so every profiled pass executes its OWN number of forwards.  Robust to that: bytes per forward
        # = sum over kernels of (mean bytes per launch, from the PMC passes) x (launches per forward, from the kernel-trace pass: calls / the
        # forwards THAT pass printed)
        def forwards_of(log):
            path = os.path.join(src, log)
            if os.path.exists(path):
                m = re.search(r'"forwards_executed": (\d+)', open(path, errors="replace").read())
                if m:
                    return int(m.group(1))
            return None
        fw_kt = forwards_of("kt.log")
        per_fw = fetch_fw = write_fw = 0.0
        if fw_kt:
            for k, v in res.items():
                if "hbm_bytes_per_launch" in v and k in stats:
                    per_launches = len(stats[k]) / fw_kt
                    per_fw += v["hbm_bytes_per_launch"] * per_launches
                    fetch_fw += v.get("hbm_fetch_bytes_corrected", 0.0) * per_launches
                    write_fw += v.get("hbm_write_bytes", 0.0) * per_launches
        if fw_kt and per_fw > 0:
            try:
                commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
            except Exception:
                commit = None
            commit = os.environ.get("GNX_PROFILE_COMMIT", commit)
            with open(os.path.join(os.path.dirname(dst) or ".", f"traffic_{key}.json"), "w") as out:
                json.dump({"__model__": {"hbm_bytes_per_launch": per_fw, "fetch_bytes_corrected": fetch_fw, "write_bytes": write_fw, "forwards_in_kernel_trace_pass": fw_kt},
                           "_meta": {"source_sha": bench.model_source_sha(core), "commit": commit, "date": datetime.date.today().isoformat(),
                                     "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --model c4`; sum over the gnx kernels of (mean bytes per launch) x (launches per forward in the kernel-trace pass); "
                                            "KiB -> B, FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH §HBM); FETCH_SIZE counts Infinity-Cache hits too"}}, out, indent=1)
    print(open(dst + "_kernel_stats.csv").read())
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
