#!/bin/bash
# README ex.3 model (enc -> 2 x GNCore(10,5,3) -> dec, 1M-edge graph): us/step and per-kernel event times per variant (environment
# switches), then a rocprofv3 kernel trace of the default.   Usage: tools/ab_c4narrow.sh ["VAR=1 ..." ...]  -> gpurun_out/ab_c4narrow/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ab_c4narrow; mkdir -p $OUT; : > $OUT/summary.txt
variants=("X=1" "$@")
for rep in 1 2; do
  for v in "${variants[@]}"; do
    line=$(env $v python3 $REPO/bench.py --model c4 --core-dims 10,5,3 --steps 50 --no-cpu-baseline 2> $OUT/err.log | tail -1)
    echo "$v rep$rep $(echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step', round(d['ms_per_step']*1e3,1), d.get('kernel_us_one_forward'))" 2>&1)" | tee -a $OUT/summary.txt
  done
done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --model c4 --core-dims 10,5,3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/kt.log 2>&1)
python3 - $OUT <<'PY'
import csv, sys, glob, collections
st = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/kt/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gnx::" in r["Kernel_Name"]:
            st[r["Kernel_Name"].split("(")[0].replace("void gnx::", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[1] + "/kernel_stats_by_instantiation.csv", "w") as o:
    o.write("kernel,calls,avg_ns,min_ns\n")
    for k, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
        o.write(f"\"{k}\",{len(v)},{sum(v)/len(v):.0f},{min(v)}\n")
print(open(sys.argv[1] + "/kernel_stats_by_instantiation.csv").read())
PY
