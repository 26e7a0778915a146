#!/bin/bash
# A/B of the core-dims (128,64,32) block on C2 under different env settings: tools/ab_wide.sh "NAME ENV=.." ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "$@"; do
  set -- $spec; name=$1; shift
  for rep in 1 2; do
    env "$@" python3 $REPO/bench.py --dims core --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$name', 'us/step', round(d['ms_per_step']*1e3,1), 'frac', d['roofline']['frac'], d['roofline']['all_kernels_us'])"
  done
done | tee -a $REPO/gpurun_out/ab_wide.log
