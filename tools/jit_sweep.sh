#!/bin/bash
# usage: tools/jit_sweep.sh "<defs A>" "<defs B>" ... -- dims1 dims2 ...   (run under gpurun)
defs=(); while [ "$1" != "--" ]; do defs+=("$1"); shift; done; shift
for d in "$@"; do for x in "${defs[@]}"; do
  r=$(GNX_JIT_ALL=1 GNX_JIT_DEFS="$x" python bench.py --dims $d --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step']*1000, r['roofline']['all_kernels_us'])" 2>&1 | tail -1)
  echo "$d [$x] $r"
done; done
