#!/bin/bash
# usage: tools/sweep.sh "ENV=.. ENV=.." ...   one bench line per env-set
for e in "$@"; do
  env $e python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e cold', d['ms_per_step'], 'warm', d['config']['warm_ms_per_step'], d['roofline']['all_kernels_us'])"
done
