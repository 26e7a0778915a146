#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
LOG=$REPO/gpurun_out/sweep_wtile.log; : > $LOG
for rep in 1 2; do
for WT in 64 128 256; do
  for cfg in "c2:" "c3:--workload hetero --hetero-graphs 512" "c5:--workload hetero --hetero-graphs 4096" "c5w:--workload hetero --hetero-graphs 4096 --hetero-edges 8000000"; do
    name=${cfg%%:*}; args=${cfg#*:}
    line=$(GNX_WTILE_E=$WT python3 $REPO/bench.py --no-cpu-baseline --no-secondary --no-c-abi --steps 100 $args 2>/dev/null | tail -1)
    echo "$name wtile_e=$WT $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("us/step", round(d["ms_per_step"]*1e3,2), "kernel_us", d["roofline"].get("kernel_us"))' 2>&1)" | tee -a $LOG
  done
done
done
