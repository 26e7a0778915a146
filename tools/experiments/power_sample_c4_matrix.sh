#!/bin/bash
# As power_sample_c4.sh, a matrix of switch settings (3000 replays each; 4 samples of power / clock from second 14 on).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
run() {
  ( env $2 timeout -k 5 60 python bench.py --model c4 --steps 3000 --warmup 5 --no-cpu-baseline --no-c-abi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_step', d['ms_per_step'])" ) &
  BP=$!
  sleep 14
  for i in 1 2 3 4; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk clock level|Socket Graphics Package Power" | sed 's/.*(\([0-9]*\)Mhz).*/\1 MHz/; s/.*Power (W): \([0-9.]*\).*/\1 W/' | tr '\n' ' '; echo
    sleep 1
  done
  wait $BP
}
run default "GNX_DUMMY=1"
run proj_fp32 "GNX_PROJ_FP32=1"
run decoder_fp32 "GNX_EDGE_NARROW_FP32=1"
run both_fp32 "GNX_PROJ_FP32=1 GNX_EDGE_NARROW_FP32=1"
run default_again "GNX_DUMMY=1"
run stats_pass "GNX_LN_STATS_PASS=1"
