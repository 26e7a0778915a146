#!/bin/bash
# The bench lines that go with a round's profiles: tools/bench_lines.sh r02  -> gpurun_out/bench_<tag>_*.json
R=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/bench_${R}_readme.json
python3 bench.py --workload hetero --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${R}_hetero512.json
python3 bench.py --workload hetero --hetero-graphs 4096 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${R}_hetero4096.json
python3 bench.py --workload hetero --hetero-graphs 4096 --hetero-edges 8000000 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${R}_hetero4096_8M.json
python3 bench.py --dims core --steps 20 2>/dev/null | tail -1 > gpurun_out/bench_${R}_core.json
python3 bench.py --dims core --workload hetero --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${R}_core_hetero512.json
python3 bench.py --model c4 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_${R}_c4.json
python3 bench.py --model c4 --core-dims 10,5,3 --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_${R}_c4narrow.json
python3 bench.py --force-dist --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${R}_forcedist.json
python3 bench.py --dims odd --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${R}_jit_odd.json
for f in gpurun_out/bench_${R}_*.json; do python3 - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d.get("roofline") or {}
print(sys.argv[1].split("bench_")[1], "value", d["value"], "ms/step", d["ms_per_step"], "frac", r.get("frac"), "whole", r.get("frac_whole_step"), "kernels", r.get("all_kernels_us", d.get("kernel_us_one_forward")))
PY
done
