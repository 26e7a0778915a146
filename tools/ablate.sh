#!/bin/bash
# timing experiments: whole-step cold/warm ms for each GNX_ABLATE mask (outputs are wrong by construction)
for m in "$@"; do
  GNX_ABLATE=$m python bench.py --no-cpu-baseline --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$m cold', d['ms_per_step'], 'warm', d['config']['warm_ms_per_step'], d['roofline']['all_kernels_us'])"
done
