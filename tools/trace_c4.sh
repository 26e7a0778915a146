#!/bin/bash
# Per-dispatch kernel trace of ONE forward of the config-4 model (Encoder -> 2 x GNCore -> Decoder): the dispatch sequence with durations
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_c4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 $REPO/bench.py --no-cpu-baseline --model c4 --steps 2 --warmup 1 "$@" > "$OUT/kt.log" 2>&1
python3 - "$OUT" <<'PY'
import sys, glob, csv
out = sys.argv[1]
f = sorted(glob.glob(out + '/kt/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# the last forward: find the last dispatch whose name starts the sequence (the first kernel name of the list after the build kernels)
import re
def short(n):
    n = re.sub(r'\(.*', '', n); n = n.replace('gnx::', '').replace('void ', '')
    return n[:70]
seq = [(short(r['Kernel_Name']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp'])) for r in rows]
# print the last 70 dispatches with gaps
# one forward = the dispatches between the last two graph-level kernels of a decoder (k_graph_final / k_skinny_dense at the very end)
for n, d, _ in seq[-70:]:
    print(f'{d:9.1f} us  {n}')
PY'
import sys, glob, csv
out = sys.argv[1]
f = sorted(glob.glob(out + '/kt/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# the last forward: find the last dispatch whose name starts the sequence (the first kernel name of the list after the build kernels)
import re
def short(n):
    n = re.sub(r'\(.*', '', n); n = n.replace('gnx::', '').replace('void ', '')
    return n[:70]
seq = [(short(r['Kernel_Name']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp'])) for r in rows]
# print the last 70 dispatches with gaps
tail = seq[-75:]
prev_end = None
tot = 0
for (n, d, s), r in zip(tail, rows[-75:]):
    gap = (s - prev_end) / 1e3 if prev_end else 0
    prev_end = int(r['End_Timestamp'])
    print(f'{d:9.1f} us  gap {gap:7.1f}  {n}')
PY
