#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
LOG=$REPO/gpurun_out/ab_proj.log; : > $LOG
for rep in 1 2 3; do
  for tag in base projhalf; do
    lib=$REPO/graphnets.jl_amd/libgnx_$tag.so
    line=$(GNX_LIB_PATH=$lib python3 $REPO/bench.py --no-cpu-baseline --no-secondary --no-c-abi --dims core --steps 20 2>/dev/null | tail -1)
    echo "core $tag $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("us/step", round(d["ms_per_step"]*1e3,2), "kernels", r.get("all_kernels_us"))' 2>&1)" | tee -a $LOG
  done
done
for tag in base projhalf; do
  lib=$REPO/graphnets.jl_amd/libgnx_$tag.so
  line=$(GNX_LIB_PATH=$lib python3 $REPO/bench.py --model c4 --no-cpu-baseline --no-c-abi --steps 10 2>/dev/null | tail -1)
  echo "c4 $tag $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms", d["ms_per_step"])' 2>&1)" | tee -a $LOG
done
