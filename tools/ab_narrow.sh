#!/bin/bash
# A/B of the fused narrow kernel's variants on the GPU box (each line = one bench.py process; values in us/step).
# Usage: tools/ab_narrow.sh [extra bench args]   -> gpurun_out/ab_narrow.log
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
LOG=$OUT/ab_narrow.log; : > $LOG
run() {  # name, env..., -- args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for rep in 1 2; do
    line=$(env "${envs[@]}" python3 $REPO/bench.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | tail -1)
    echo "$name rep$rep $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("us/step", round(d["ms_per_step"]*1e3,2), "warm", round(d["config"].get("warm_ms_per_step",0)*1e3,2), "kernels", r["all_kernels_us"], "frac", r["frac"], "whole", r["frac_whole_step"])' 2>&1)" | tee -a $LOG
  done
}
run c2 X=1 -- "$@"
run hetero512 X=1 -- --workload hetero "$@"
run hetero4096 X=1 -- --workload hetero --hetero-graphs 4096 "$@"
