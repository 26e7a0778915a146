#!/bin/bash
# round-3 GPU check: the whole GPU suite, smoke, then the driver's bench invocation
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r3_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
S0=$(date +%s); timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - S0 )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_whole_step"], d["roofline"]["traffic"], d["roofline"].get("traffic_breakdown"), d["batch_ms"], d.get("pipelined_two_streams"))
for k,v in d["secondary"].items():
    if isinstance(v, dict):
        print(k, {x: v.get(x) for x in ("value","ms_per_step","wall_s","error","batch_ms")}, v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("traffic"), (v.get("cpu_baseline") or {}).get("value"), (v.get("pipelined_two_streams") or {}).get("ms_per_step"))
    else: print(k, v)
print(len(json.dumps(d)), "bytes")
PY
