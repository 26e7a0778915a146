#!/bin/bash
# Per-dispatch kernel trace of the config-4 model (Encoder -> 2 x GNCore -> Decoder): the last dispatches with their durations.
# (The kernels a wide GNCore puts on its side stream overlap others: GNX_NO_FORK=1 bash tools/trace_c4.sh for undisturbed durations.)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_c4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 $REPO/bench.py --no-cpu-baseline --model c4 --steps 2 --warmup 1 "$@" > "$OUT/kt.log" 2>&1
python3 - "$OUT" <<'PY'
import sys, glob, csv, os, re
out = sys.argv[1]
f = max(glob.glob(out + '/kt/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\(.*', '', n)
    return n.replace('gnx::', '').replace('void ', '')[:70]
for r in rows[-70:]:
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  {short(r['Kernel_Name'])}")
PY
