#!/bin/bash
# Socket power and shader clock (rocm-smi, every second) while bench.py replays the config-4 forward 3000 times, for three builds-by-switch of the same library:
# default; GNX_PROJ_FP32=1 GNX_EDGE_NARROW_FP32=1 (110 us more kernel time per forward, same model time on most boxes); GNX_CORE_EDGE_SPLIT=1 (two launches per core).
# Usage: bash tools/experiments/power_sample_c4.sh > gpurun_out/power_sample_c4_variants.log
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
run() {
  echo "== $1"
  ( env $2 timeout -k 5 60 python bench.py --model c4 --steps 3000 --warmup 5 --no-cpu-baseline --no-c-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])" ) &
  BP=$!
  sleep 12
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk clock level|Socket Graphics Package Power" | tr '\n' ' '; echo
    sleep 1
  done
  wait $BP
}
run default "GNX_DUMMY=1"
run proj_and_decoder_fp32 "GNX_PROJ_FP32=1 GNX_EDGE_NARROW_FP32=1"
run two_launches_per_core "GNX_CORE_EDGE_SPLIT=1"
