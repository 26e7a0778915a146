#!/bin/bash
# Same-box A/B of library builds: tools/experiments/ab_libs.sh <tag> [<tag> ...]   (tag "main" = graphnets.jl_amd/libgnx.so, else libgnx_<tag>.so;
# variants from tools/build_variant.sh).  Two rounds, alternating, of: one captured GNCore forward (core_replay_time.py) and config 4 (bench.py --model c4).
# Devices differ by up to 12 % on matrix-bound kernels and config 4 has two clock states: only figures from ONE gpurun call compare.
for round in 1 2; do
  for tag in "$@"; do
    if [ "$tag" = main ]; then unset GNX_LIB_PATH; else export GNX_LIB_PATH=graphnets.jl_amd/libgnx_$tag.so; fi
    core=$(python tools/experiments/core_replay_time.py 2>/dev/null | tail -1 | cut -d' ' -f1)
    c4=$(python bench.py --model c4 --no-cpu-baseline --no-c-abi --full-line --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_us_one_forward'].get('k_core_edge_x6'))")
    echo "round $round  $tag: GNCore $core ms   config 4 $c4"
  done
done
