#!/bin/bash
# Per-dispatch kernel trace of the config-4 model (Encoder -> 2 x GNCore -> Decoder): the last dispatches with their durations.
# (The kernels a wide GNCore puts on its side stream overlap others: GNX_NO_FORK=1 bash tools/trace_c4.sh for undisturbed durations.)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_c4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 $REPO/bench.py --no-cpu-baseline --no-c-abi --model c4 --steps 2 --warmup 1 "$@" > "$OUT/kt.log" 2>&1
python3 - "$OUT" <<'PY'
import sys, glob, csv, os, re
out = sys.argv[1]
f = max(glob.glob(out + '/kt/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\(.*', '', n)
    return n.replace('gnx::', '').replace('void ', '')[:70]
t0 = int(rows[-70]['Start_Timestamp']) if len(rows) >= 70 else int(rows[0]['Start_Timestamp'])
end_prev = t0
for r in rows[-70:]:  # start relative to the first listed dispatch, gap since the latest end so far (negative: overlap), duration
    b, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(b - t0) / 1e3:10.1f}  gap {(b - end_prev) / 1e3:8.1f}  {(e - b) / 1e3:9.1f} us  {short(r['Kernel_Name'])}")
    end_prev = max(end_prev, e)
PY
