#!/bin/bash
# the driver's invocation (headline + secondary configs in ONE line) and two extra lines, kept under profiles/
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r03_default.json 2> gpurun_out/bench_r03_default.err; echo "default rc=$?"
timeout -k 10 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-secondary > gpurun_out/bench_r03_readme_200.json 2>/dev/null; echo "readme200 rc=$?"
timeout -k 10 300 python bench.py --force-dist --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r03_forcedist.json 2>/dev/null; echo "forcedist rc=$?"
python - <<'PY'
import json
for f in ("bench_r03_default","bench_r03_readme_200","bench_r03_forcedist"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("frac_whole_step"), r.get("traffic"), r.get("kernel_us"), (d.get("pipelined_two_streams") or {}).get("ms_per_step"), (d.get("cpu_baseline") or {}).get("value"))
    for k,v in (d.get("secondary") or {}).items():
        if isinstance(v, dict): print("   ", k, v.get("value"), v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("frac_whole_step"), v.get("roofline",{}).get("traffic"), (v.get("cpu_baseline") or {}).get("value"), v.get("batch_ms",{}) and {a:b for a,b in v["batch_ms"].items() if a!="what"}, (v.get("pipelined_two_streams") or {}).get("ms_per_step"))
PY
