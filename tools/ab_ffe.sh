#!/bin/bash
# A/B on the GPU box: README ex.3 at its own widths with the edge FeedForward in the block kernel's edge lanes (default) and with the
# two-kernel form (GNX_NO_FFE=1); alternating, three pairs.  -> gpurun_out/ab_ffe.log
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
for i in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export GNX_NO_FFE=1; else unset GNX_NO_FFE; fi
    python bench.py --model c4 --core-dims 10,5,3 --steps 50 --warmup 5 --no-cpu-baseline --no-c-abi 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NO_FFE=$v', 'ms/step', d['ms_per_step'], 'kernels', d['kernel_us_one_forward'])"
  done
done | tee gpurun_out/ab_ffe.log
