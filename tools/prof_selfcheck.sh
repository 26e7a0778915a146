#!/bin/bash
# gnx_profile_* (dispatch timestamps) against rocprofv3's kernel trace on the SAME launches: bench.py's per-kernel pass runs under the
# profiler, its JSON line carries the library's figures, the trace the profiler's.  Usage (GPU box): tools/prof_selfcheck.sh <tag> [bench args]
# -> gpurun_out/selfcheck_<tag>.txt (kernel, library avg us, rocprofv3 avg us over the launches of the per-kernel pass, ratio)
set -u
TAG=${1:-readme}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/selfcheck_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 $REPO/bench.py --no-cpu-baseline --no-secondary --no-c-abi "$@" > "$OUT/kt.log" 2>&1
python3 $REPO/tools/prof_selfcheck.py "$OUT" | tee $REPO/gpurun_out/selfcheck_$TAG.txt
