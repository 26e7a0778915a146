#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and, in SEPARATE passes, the HBM traffic counters of bench.py.
# Usage: tools/profile.sh <tag> [bench args...]   → gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
