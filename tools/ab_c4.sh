#!/bin/bash
# A/B of the C4 model (BASELINE configs[3]) under different env settings: tools/ab_c4.sh "NAME ENV=.. ENV=.." ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "$@"; do
  set -- $spec; name=$1; shift
  for rep in 1 2; do
    env "$@" python3 $REPO/bench.py --model c4 --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
k=d['kernel_us_one_forward']
print('$name', 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], {n:v for n,v in k.items() if 'ffn' in n or 'gemm_edge' in n or 'layernorm' in n})"
  done
done | tee -a $REPO/gpurun_out/ab_c4.log
